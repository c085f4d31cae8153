"""Div2K / SetXX super-resolution data provider -- the loader side of the hot path
(reference ofa/imagenet_codebase/data_providers/div2k_setxx.py:14-225 provider, :288-298 dataset, :300-379 PIL
transforms; base_provider.py:13-131 DataProvider / MyDistributedSampler).

Same contract as the reference: every sample is a dict
    {'image': HR, '2x_down_image': PIL-bicubic HR/2, '4x_down_image': PIL-bicubic HR/4}
of float32 CHW tensors in [0, 1]; training images are RandomCrop(image_size) -> RandomHorizontalFlip ->
RandomRotation((-90, 90)) of the decoded RGB image, validation / test images are ModCrop(4) full images
(batch 1).  The reference builds its augmentation from torchvision.transforms; torchvision is not part of this
image, so the three random transforms are restated here on PIL with the same parameter draws from the torch
global RNG, in the same order (torchvision: RandomCrop.get_params -> two torch.randint, RandomHorizontalFlip ->
torch.rand(1) < p, RandomRotation.get_params -> torch.empty(1).uniform_(lo, hi); rotation is PIL
Image.rotate(angle, NEAREST, expand=False), fill 0).  ModCrop / Scale / get_transform_L / the dataset / the
samplers are plain PIL + torch in the reference too and are pinned by tests/golden/div2k.npz.
"""
import os
import warnings

import numpy as np
import torch
import torch.utils.data

try:   # PIL is optional at import time so that the GPU-side modules never depend on it
    from PIL import Image
except ImportError:   # pragma: no cover
    Image = None

IMG_EXTENSIONS = [".jpg", ".JPG", ".jpeg", ".JPEG", ".png", ".PNG", ".ppm", ".PPM", ".bmp", ".BMP"]


def is_image_file(filename):
    return any(filename.endswith(ext) for ext in IMG_EXTENSIONS)


def get_image_paths_recursive(dir, images):
    """reference :270-280 -- os.walk already descends, and every sub-directory is walked again on top of that, so
    images below the root are listed once per ancestor: kept, because it defines the epoch length and the index ->
    file map of a nested tree."""
    assert os.path.isdir(dir), "%s is not a valid directory" % dir
    for root, subdirs, fnames in sorted(os.walk(dir)):
        for fname in fnames:
            if is_image_file(fname):
                images.append(os.path.join(root, fname))
        for subdir in subdirs:
            get_image_paths_recursive(os.path.join(root, subdir), images)
    return images


# ------------------------------------------------------------------------------------------- transforms
def crop(img, i, j, h, w):
    return img.crop((j, i, j + w, i + h))


class ModCrop(object):
    """crop (top-left anchored) so both sides are divisible by `mod` (reference :314-344)"""

    def __init__(self, mod):
        self.mod = int(mod)

    @staticmethod
    def get_params(img, mod):
        w, h = img.size
        return 0, 0, h - h % mod, w - w % mod

    def __call__(self, img):
        i, j, h, w = self.get_params(img, self.mod)
        return crop(img, i, j, h, w)


def scale(img, size, interpolation=None):
    assert isinstance(size, tuple) and len(size) == 2
    return img.resize(size[::-1], Image.BICUBIC if interpolation is None else interpolation)   # (h, w) -> (w, h)


class Scale(object):
    """PIL resize by a scale factor, output size int(w*s) x int(h*s) (reference :354-368)"""

    def __init__(self, scale_factor, interpolation=None):
        self.scale_factor = scale_factor
        self.interpolation = interpolation

    @staticmethod
    def get_params(img, scale_factor):
        w, h = img.size
        return int(h * scale_factor), int(w * scale_factor)

    def __call__(self, img):
        return scale(img, self.get_params(img, self.scale_factor), self.interpolation)


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, img):
        for t in self.transforms:
            img = t(img)
        return img


def get_transform_L(opt=4):
    """HR -> LR by 1/opt, PIL bicubic (reference :370-379)"""
    assert opt in (2, 4, 8)
    return Compose([Scale(scale_factor=1 / opt)])


class RandomCrop(object):
    """torchvision.transforms.RandomCrop(size) without padding; offsets drawn like torchvision's get_params"""

    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, (int, float)) else tuple(size)

    def __call__(self, img):
        w, h = img.size
        th, tw = self.size
        if h < th or w < tw:
            raise ValueError("Required crop size %s is larger than input image size %s" % ((th, tw), (h, w)))
        if w == tw and h == th:
            return img
        i = int(torch.randint(0, h - th + 1, size=(1,)).item())
        j = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        return crop(img, i, j, th, tw)


class RandomHorizontalFlip(object):
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, img):
        if torch.rand(1) < self.p:
            return img.transpose(Image.FLIP_LEFT_RIGHT)
        return img


class RandomRotation(object):
    """torchvision.transforms.RandomRotation(degrees): nearest resampling, same canvas, zero fill"""

    def __init__(self, degrees):
        self.degrees = (-float(degrees), float(degrees)) if isinstance(degrees, (int, float)) else \
            (float(degrees[0]), float(degrees[1]))

    def __call__(self, img):
        angle = float(torch.empty(1).uniform_(self.degrees[0], self.degrees[1]).item())
        return img.rotate(angle, Image.NEAREST, False, None)


def to_tensor(img):
    """torchvision ToTensor for an 8-bit PIL image: HWC uint8 -> CHW float32 / 255"""
    a = np.asarray(img, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1))).float().div_(255.0)


# ---------------------------------------------------------------------------------------------- dataset
class Div2K_SetXXDataset(torch.utils.data.Dataset):
    """reference :288-298.  `samples` aliases `paths` (the reference's provider reads `dataset.samples`, an
    ImageFolder attribute its own dataset class does not define -- :51,57,206)."""

    def __init__(self, root_dir, transform=None, lr_on_device=False):
        self.root_dir = root_dir
        self.transform = transform
        self.lr_on_device = lr_on_device   # yield only the uint8 HR image; the LR images are made on the GPU
        self.paths = get_image_paths_recursive(self.root_dir, [])
        self.samples = self.paths
        self.size = len(self.paths)

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        img = Image.open(self.paths[index]).convert("RGB")
        H_img = self.transform(img) if self.transform is not None else img
        if self.lr_on_device:   # 1/4 of the float bytes cross PCIe; ops.lr_images_from_u8 does the PIL bicubic on the GPU
            a = np.asarray(H_img, dtype=np.uint8)
            return {"image_u8": torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))}
        L2_img = get_transform_L(opt=2)(H_img)
        L4_img = get_transform_L(opt=4)(H_img)
        return {"image": to_tensor(H_img), "2x_down_image": to_tensor(L2_img), "4x_down_image": to_tensor(L4_img)}


class RankShardSampler(torch.utils.data.Sampler):
    """One rank's share of an index set: a permutation seeded by the epoch (identical on every rank), padded by
    wrapping to a multiple of the world size, then every `num_replicas`-th index starting at `rank` -- the
    partition rule of the reference's MyDistributedSampler (base_provider.py:106-131) written as a plain Sampler.
    Call set_epoch(e) before each epoch or every epoch replays the same order."""

    def __init__(self, indices, num_replicas, rank, shuffle=True):
        self.indices = np.asarray(indices, dtype=np.int64)
        self.num_replicas, self.rank, self.shuffle = int(num_replicas), int(rank), bool(shuffle)
        if not 0 <= self.rank < self.num_replicas:
            raise ValueError("rank %d outside a world of %d" % (self.rank, self.num_replicas))
        self.num_samples = -(-len(self.indices) // self.num_replicas)
        self.total_size = self.num_samples * self.num_replicas
        self.epoch = 0

    def set_epoch(self, epoch):
        self.epoch = int(epoch)

    def __len__(self):
        return self.num_samples

    def __iter__(self):
        n = len(self.indices)
        if self.shuffle:
            order = torch.randperm(n, generator=torch.Generator().manual_seed(self.epoch)).numpy()
        else:
            order = np.arange(n)
        order = np.concatenate([order, order[:self.total_size - n]])
        return iter(self.indices[order[self.rank::self.num_replicas]].tolist())


class MyDistributedSampler(RankShardSampler):
    """the reference's constructor signature (dataset, num_replicas, rank, sub_index_list) on RankShardSampler"""

    def __init__(self, dataset, num_replicas=None, rank=None, sub_index_list=None):
        idx = np.arange(len(dataset)) if sub_index_list is None else sub_index_list
        super().__init__(idx, num_replicas, rank)
        self.sub_index_list = self.indices


# --------------------------------------------------------------------------------------------- provider
class DataProvider(object):
    SUB_SEED = 937162211     # random seed for sampling subset
    VALID_SEED = 2147483647  # random seed for the validation set

    @staticmethod
    def random_sample_valid_set(train_size, valid_size):
        assert train_size > valid_size
        perm = torch.randperm(train_size, generator=torch.Generator().manual_seed(DataProvider.VALID_SEED)).tolist()
        return perm[valid_size:], perm[:valid_size]


class Div2K_SetXXDataProvider(DataProvider):
    """DIV2K training crops + SetXX full-size validation images (interface of the reference's provider,
    div2k_setxx.py:17-225: `train` / `valid` / `test` loaders, `data_shape`, `assign_active_img_size`,
    `build_sub_train_loader`).

    Sharding under data parallelism (num_replicas given): the TRAINING indices are split over the ranks by
    RankShardSampler; the validation / test images are NOT split -- every rank evaluates the whole (14-image) set in
    file order, so the logged PSNR, `best_acc` and `model_best` are the same on every rank count.  (The reference
    wraps the test set in a shuffling, padding DistributedSampler with drop_last and never reduces the metric.)"""
    DEFAULT_PATH = "/SSD/div2k_setxx"

    def __init__(self, save_path=None, train_batch_size=256, test_batch_size=512, valid_size=None, n_worker=32,
                 resize_scale=0.08, distort_color=None, image_size=32, num_replicas=None, rank=None,
                 lr_on_device=False):
        warnings.filterwarnings("ignore")
        self.lr_on_device = bool(lr_on_device)
        if Image is None:
            raise ImportError("Div2K_SetXXDataProvider needs PIL")
        self._save_path = save_path
        self.distort_color, self.resize_scale = distort_color, resize_scale
        self._workers = n_worker
        # a list of sizes selects multi-size training in the classification providers; the SR transform list has the
        # resizing crop commented out (reference :167-176), so only the largest size is ever active
        sizes = sorted(image_size) if isinstance(image_size, (list, tuple)) else [image_size]
        self.image_size = image_size if isinstance(image_size, int) else sizes
        self.active_img_size = sizes[-1]
        self._valid_transform_dict = {sz: self.build_valid_transform(sz) for sz in sizes}
        eval_tf = self._valid_transform_dict[self.active_img_size]

        train_set = self.train_dataset(self.build_train_transform())
        sharded = num_replicas is not None and num_replicas > 1
        n_train = len(train_set.samples)
        train_idx, valid_idx = list(range(n_train)), None
        if valid_size is not None:
            if isinstance(valid_size, float):
                assert 0 < valid_size < 1
                valid_size = int(n_train * valid_size)
            train_idx, valid_idx = self.random_sample_valid_set(n_train, valid_size)
        if sharded:
            train_sampler = RankShardSampler(train_idx, num_replicas, rank)
        elif valid_idx is not None:
            train_sampler = torch.utils.data.SubsetRandomSampler(train_idx)
        else:
            train_sampler = torch.utils.data.RandomSampler(train_set)
        self.train = self._loader(train_set, train_batch_size, train_sampler, drop_last=True)
        self.test = self._loader(self.test_dataset(eval_tf), test_batch_size, None, drop_last=False)
        if valid_idx is None:
            self.valid = self.test
        else:   # held-out training images, full set on every rank, fixed order
            held_out = torch.utils.data.Subset(self.train_dataset(eval_tf), sorted(valid_idx))
            self.valid = self._loader(held_out, test_batch_size, None, drop_last=False)

    def _loader(self, dataset, batch_size, sampler, drop_last):
        return torch.utils.data.DataLoader(dataset, batch_size=batch_size, sampler=sampler, shuffle=False,
                                           num_workers=self._workers, pin_memory=True, drop_last=drop_last)

    @staticmethod
    def name():
        return "div2k_setxx"

    @property
    def data_shape(self):
        return 3, self.active_img_size, self.active_img_size

    @property
    def n_classes(self):
        return 1

    @property
    def save_path(self):
        return self._save_path or self.DEFAULT_PATH

    @property
    def data_url(self):
        raise ValueError("unable to download %s" % self.name())

    @property
    def train_path(self):
        return os.path.join(self.save_path, "train")

    @property
    def valid_path(self):
        return os.path.join(self.save_path, "val")

    def train_dataset(self, _transforms):
        return Div2K_SetXXDataset(self.train_path, _transforms, lr_on_device=getattr(self, "lr_on_device", False))

    def test_dataset(self, _transforms):
        return Div2K_SetXXDataset(self.valid_path, _transforms, lr_on_device=getattr(self, "lr_on_device", False))

    def build_train_transform(self, image_size=None, print_log=True):
        """RandomCrop -> RandomHorizontalFlip -> RandomRotation((-90, 90)) (reference :145-180; its resizing crop and
        colour jitter are commented out / unused by the SR scripts, which pass distort_color=None)"""
        if self.distort_color not in (None, "None"):
            raise NotImplementedError("distort_color=%r: colour jitter needs torchvision, which this image lacks; the "
                                      "SR entry scripts train with distort_color=None" % (self.distort_color,))
        side = image_size if image_size is not None else self.active_img_size
        return Compose([RandomCrop(side), RandomHorizontalFlip(), RandomRotation(degrees=(-90, 90))])

    def build_valid_transform(self, image_size=None):
        return Compose([ModCrop(mod=4)])

    def _eval_datasets(self):
        for loader in {id(self.valid): self.valid, id(self.test): self.test}.values():
            ds = loader.dataset
            yield ds.dataset if isinstance(ds, torch.utils.data.Subset) else ds

    def assign_active_img_size(self, new_img_size):
        self.active_img_size = new_img_size
        tf = self._valid_transform_dict.setdefault(new_img_size, self.build_valid_transform(new_img_size))
        for ds in self._eval_datasets():
            ds.transform = tf

    def build_sub_train_loader(self, n_images, batch_size, num_worker=None, num_replicas=None, rank=None):
        """`n_images` fixed training images (seed SUB_SEED) as a cached list of batches, for BN re-calibration
        (reference :199-225).  The batches stay the dicts the dataset yields -- what set_running_statistics reads."""
        cache = self.__dict__.setdefault("_sub_train_cache", {})
        key = (self.active_img_size, n_images, batch_size)
        if key not in cache:
            ds = self.train_dataset(self.build_train_transform(image_size=self.active_img_size, print_log=False))
            perm = torch.randperm(len(ds.samples), generator=torch.Generator().manual_seed(DataProvider.SUB_SEED))
            chosen = perm[:n_images].tolist()
            if num_replicas is not None and num_replicas > 1:
                sampler = RankShardSampler(chosen, num_replicas, rank)
            else:
                sampler = torch.utils.data.SubsetRandomSampler(chosen)
            workers = self._workers if num_worker is None else num_worker
            cache[key] = list(torch.utils.data.DataLoader(ds, batch_size=batch_size, sampler=sampler,
                                                          num_workers=workers, pin_memory=True))
        return cache[key]
