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
import math
import os
import warnings

import numpy as np
import torch
import torch.utils.data
from torch.utils.data.distributed import DistributedSampler

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

    def __init__(self, root_dir, transform=None):
        self.root_dir = root_dir
        self.transform = transform
        self.paths = get_image_paths_recursive(self.root_dir, [])
        self.samples = self.paths
        self.size = len(self.paths)

    def __len__(self):
        return self.size

    def __getitem__(self, index):
        img = Image.open(self.paths[index]).convert("RGB")
        H_img = self.transform(img) if self.transform is not None else img
        L2_img = get_transform_L(opt=2)(H_img)
        L4_img = get_transform_L(opt=4)(H_img)
        return {"image": to_tensor(H_img), "2x_down_image": to_tensor(L2_img), "4x_down_image": to_tensor(L4_img)}


class MyDistributedSampler(DistributedSampler):
    """DistributedSampler over a subset of indices (reference base_provider.py:106-131)"""

    def __init__(self, dataset, num_replicas=None, rank=None, sub_index_list=None):
        super(MyDistributedSampler, self).__init__(dataset, num_replicas, rank)
        self.sub_index_list = sub_index_list   # numpy
        self.num_samples = int(math.ceil(len(self.sub_index_list) * 1.0 / self.num_replicas))
        self.total_size = self.num_samples * self.num_replicas

    def __iter__(self):
        g = torch.Generator()
        g.manual_seed(self.epoch)
        indices = torch.randperm(len(self.sub_index_list), generator=g).tolist()
        indices += indices[:(self.total_size - len(indices))]
        indices = self.sub_index_list[indices].tolist()
        assert len(indices) == self.total_size
        indices = indices[self.rank:self.total_size:self.num_replicas]
        assert len(indices) == self.num_samples
        return iter(indices)


# --------------------------------------------------------------------------------------------- provider
class DataProvider(object):
    SUB_SEED = 937162211     # random seed for sampling subset
    VALID_SEED = 2147483647  # random seed for the validation set

    @staticmethod
    def random_sample_valid_set(train_size, valid_size):
        assert train_size > valid_size
        g = torch.Generator()
        g.manual_seed(DataProvider.VALID_SEED)
        rand_indexes = torch.randperm(train_size, generator=g).tolist()
        return rand_indexes[valid_size:], rand_indexes[:valid_size]


class Div2K_SetXXDataProvider(DataProvider):
    DEFAULT_PATH = "/SSD/div2k_setxx"

    def __init__(self, save_path=None, train_batch_size=256, test_batch_size=512, valid_size=None, n_worker=32,
                 resize_scale=0.08, distort_color=None, image_size=32, num_replicas=None, rank=None):
        warnings.filterwarnings("ignore")
        if Image is None:
            raise ImportError("Div2K_SetXXDataProvider needs PIL")
        self._save_path = save_path
        self.image_size = image_size
        self.distort_color = distort_color
        self.resize_scale = resize_scale
        self._valid_transform_dict = {}
        if not isinstance(self.image_size, int):
            # the reference switches to MyDataLoader + MyRandomResizedCrop for multi-size training, but its SR
            # transform list has that crop commented out (:167-176): only the largest size is ever active
            assert isinstance(self.image_size, list)
            self.image_size.sort()
            for img_size in self.image_size:
                self._valid_transform_dict[img_size] = self.build_valid_transform(img_size)
            self.active_img_size = max(self.image_size)
            valid_transforms = self._valid_transform_dict[self.active_img_size]
        else:
            self.active_img_size = self.image_size
            valid_transforms = self.build_valid_transform()
        loader = torch.utils.data.DataLoader

        train_transforms = self.build_train_transform()
        train_dataset = self.train_dataset(train_transforms)
        if valid_size is not None:
            if not isinstance(valid_size, int):
                assert isinstance(valid_size, float) and 0 < valid_size < 1
                valid_size = int(len(train_dataset.samples) * valid_size)
            valid_dataset = self.train_dataset(valid_transforms)
            train_indexes, valid_indexes = self.random_sample_valid_set(len(train_dataset.samples), valid_size)
            if num_replicas is not None:
                train_sampler = MyDistributedSampler(train_dataset, num_replicas, rank, np.array(train_indexes))
                valid_sampler = MyDistributedSampler(valid_dataset, num_replicas, rank, np.array(valid_indexes))
            else:
                train_sampler = torch.utils.data.sampler.SubsetRandomSampler(train_indexes)
                valid_sampler = torch.utils.data.sampler.SubsetRandomSampler(valid_indexes)
            self.train = loader(train_dataset, batch_size=train_batch_size, sampler=train_sampler,
                                num_workers=n_worker, pin_memory=True, drop_last=True)
            self.valid = loader(valid_dataset, batch_size=test_batch_size, sampler=valid_sampler,
                                num_workers=n_worker, pin_memory=True, drop_last=True)
        else:
            if num_replicas is not None:
                train_sampler = DistributedSampler(train_dataset, num_replicas, rank)
                self.train = loader(train_dataset, batch_size=train_batch_size, sampler=train_sampler,
                                    num_workers=n_worker, pin_memory=True, drop_last=True)
            else:
                self.train = loader(train_dataset, batch_size=train_batch_size, shuffle=True, num_workers=n_worker,
                                    pin_memory=True, drop_last=True)
            self.valid = None

        test_dataset = self.test_dataset(valid_transforms)
        if num_replicas is not None:
            test_sampler = DistributedSampler(test_dataset, num_replicas, rank)
            self.test = loader(test_dataset, batch_size=test_batch_size, sampler=test_sampler, num_workers=n_worker,
                               pin_memory=True, drop_last=True)
        else:
            self.test = loader(test_dataset, batch_size=test_batch_size, shuffle=True, num_workers=n_worker,
                               pin_memory=True, drop_last=True)
        if self.valid is None:
            self.valid = self.test

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
        if self._save_path is None:
            self._save_path = self.DEFAULT_PATH
        return self._save_path

    @property
    def data_url(self):
        raise ValueError("unable to download %s" % self.name())

    def train_dataset(self, _transforms):
        return Div2K_SetXXDataset(self.train_path, _transforms)

    def test_dataset(self, _transforms):
        return Div2K_SetXXDataset(self.valid_path, _transforms)

    @property
    def train_path(self):
        return os.path.join(self.save_path, "train")

    @property
    def valid_path(self):
        return os.path.join(self.save_path, "val")

    def build_train_transform(self, image_size=None, print_log=True):
        if image_size is None:
            image_size = self.image_size if isinstance(self.image_size, int) else max(self.image_size)
        if self.distort_color in ("torch", "tf"):
            raise NotImplementedError("ColorJitter needs torchvision (the SR scripts run with distort_color=None)")
        return Compose([RandomCrop(image_size), RandomHorizontalFlip(), RandomRotation(degrees=(-90, 90))])

    def build_valid_transform(self, image_size=None):
        return Compose([ModCrop(mod=4)])

    def assign_active_img_size(self, new_img_size):
        self.active_img_size = new_img_size
        if self.active_img_size not in self._valid_transform_dict:
            self._valid_transform_dict[self.active_img_size] = self.build_valid_transform()
        self.valid.dataset.transform = self._valid_transform_dict[self.active_img_size]
        self.test.dataset.transform = self._valid_transform_dict[self.active_img_size]

    def build_sub_train_loader(self, n_images, batch_size, num_worker=None, num_replicas=None, rank=None):
        """loader of `n_images` fixed training images for BN re-calibration.  The reference caches a list of
        `(images, labels)` tuples (:222-224), which its dict-yielding dataset cannot produce; the batches are kept
        as the dicts the dataset yields (what set_running_statistics reads)."""
        key = "sub_train_%d" % self.active_img_size
        if self.__dict__.get(key, None) is None:
            if num_worker is None:
                num_worker = self.train.num_workers
            n_samples = len(self.train.dataset.samples)
            g = torch.Generator()
            g.manual_seed(DataProvider.SUB_SEED)
            rand_indexes = torch.randperm(n_samples, generator=g).tolist()
            new_train_dataset = self.train_dataset(
                self.build_train_transform(image_size=self.active_img_size, print_log=False))
            chosen_indexes = rand_indexes[:n_images]
            if num_replicas is not None:
                sub_sampler = MyDistributedSampler(new_train_dataset, num_replicas, rank, np.array(chosen_indexes))
            else:
                sub_sampler = torch.utils.data.sampler.SubsetRandomSampler(chosen_indexes)
            sub_data_loader = torch.utils.data.DataLoader(new_train_dataset, batch_size=batch_size, sampler=sub_sampler,
                                                          num_workers=num_worker, pin_memory=True)
            self.__dict__[key] = [batch for batch in sub_data_loader]
        return self.__dict__[key]
