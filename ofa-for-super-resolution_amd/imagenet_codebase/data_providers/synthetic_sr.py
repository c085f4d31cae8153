"""Synthetic SR data provider: the loader-side contract of the reference's Div2K_SetXX provider
(ofa/imagenet_codebase/data_providers/div2k_setxx.py:17-225,288-298) without the dataset.

Every batch is a dict {'image': HR, '2x_down_image': HR/2, '4x_down_image': HR/4}, float32 NCHW in
[0,1] -- exactly what progressive_shrinking.train_one_epoch and SRRunManager.validate consume.  The
real provider needs torchvision + /SSD/div2k_setxx (absent here); this one
makes seeded images and bicubic(antialias) down-scales them, which is the shape- and
range-faithful stand-in used by the tests and by bench.py.
"""
import torch
import torch.nn.functional as F


class _ListLoader(object):
    """a re-iterable, len()-able list of dict batches (what the training loops need of a DataLoader)."""

    def __init__(self, batches):
        self.batches = list(batches)

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def make_batch(batch_size, hr_size, generator=None, device="cpu"):
    if isinstance(hr_size, int):
        hr_size = (hr_size, hr_size)
    hr = torch.rand((batch_size, 3, hr_size[0], hr_size[1]), generator=generator)
    x2 = F.interpolate(hr, scale_factor=0.5, mode="bicubic", antialias=True).clamp_(0, 1)
    x4 = F.interpolate(hr, scale_factor=0.25, mode="bicubic", antialias=True).clamp_(0, 1)
    return {"image": hr.to(device), "2x_down_image": x2.to(device), "4x_down_image": x4.to(device)}


class SyntheticSRDataProvider(object):
    """attributes mirror DataProvider (base_provider.py): data_shape, image_size, train/valid/test."""

    def __init__(self, train_batch_size=16, test_batch_size=1, image_size=256, n_train_batches=4,
                 n_test_batches=2, seed=0, rank=0, num_replicas=1, device="cpu", test_sizes=None):
        self.image_size = image_size
        self.active_img_size = image_size
        self.n_classes = None
        g = torch.Generator().manual_seed(seed * 1000003 + rank)
        self.train = _ListLoader(make_batch(train_batch_size, image_size, g, device) for _ in range(n_train_batches))
        gt = torch.Generator().manual_seed(seed * 1000003 + 7919)
        sizes = test_sizes or [image_size] * n_test_batches
        # validation runs batch-1 full images whose sides are multiples of 4 (ModCrop(4), div2k_setxx.py:182-190)
        self.test = _ListLoader(make_batch(test_batch_size, s, gt, device) for s in sizes)
        self.valid = self.test

    @staticmethod
    def name():
        return "synthetic_sr"

    @property
    def data_shape(self):
        return 3, self.active_img_size, self.active_img_size

    def build_sub_train_loader(self, n_images, batch_size, num_worker=None, num_replicas=None, rank=None):
        return self.train
