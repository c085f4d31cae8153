"""Host-side helpers of the hot path: activation factory (PixelShuffle / PixelUnshuffle on the HIP
kernels), shape helpers, the PSNR metric and the MyModule / MyNetwork bases.

Mirror of the parts of the reference's ofa/utils.py that the SR path uses (file:line cited per
symbol).  Classification-only helpers (accuracy, Hswish, SE, download_url) are out of scope.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from . import ops


# ------------------------------------------------------------------------------------ shapes
def get_same_padding(kernel_size):
    """reference ofa/utils.py:211-219"""
    if isinstance(kernel_size, tuple):
        assert len(kernel_size) == 2, "invalid kernel size: %s" % (kernel_size,)
        return get_same_padding(kernel_size[0]), get_same_padding(kernel_size[1])
    assert isinstance(kernel_size, int), "kernel size should be either `int` or `tuple`"
    assert kernel_size % 2 > 0, "kernel size should be odd number"
    return kernel_size // 2


def make_divisible(v, divisor, min_val=None):
    """reference ofa/utils.py:222-239 (round to a multiple of `divisor`, never below 90 %)."""
    floor = divisor if min_val is None else min_val
    out = max(floor, int(v + divisor / 2) // divisor * divisor)
    return out + divisor if out < 0.9 * v else out


def sub_filter_start_end(kernel_size, sub_kernel_size):
    """centre-crop window (reference ofa/imagenet_codebase/utils/__init__.py:89-94)."""
    start = kernel_size // 2 - sub_kernel_size // 2
    end = start + sub_kernel_size
    assert end - start == sub_kernel_size
    return start, end


def int2list(val, repeat_time=1):
    """reference ofa/imagenet_codebase/utils/__init__.py:97-103 -- NB a list is returned as is
    (callers rely on / suffer from the aliasing)."""
    if isinstance(val, (list, np.ndarray)):
        return val
    if isinstance(val, tuple):
        return list(val)
    return [val for _ in range(repeat_time)]


def list_mean(x):
    return sum(x) / len(x)


def subset_mean(val_list, sub_indexes):
    sub_indexes = int2list(sub_indexes, 1)
    return list_mean([val_list[idx] for idx in sub_indexes])


def get_net_device(net):
    return next(net.parameters()).device


# ------------------------------------------------------------------------------- activations
class PixelShuffle(nn.Module):
    """nn.PixelShuffle drop-in on the HIP sub-pixel reshuffle kernel (reference ofa/utils.py:309-310)."""

    def __init__(self, upscale_factor=2):
        super().__init__()
        self.upscale_factor = upscale_factor

    def forward(self, x):
        return ops.pixel_shuffle(x, self.upscale_factor)

    def extra_repr(self):
        return "upscale_factor=%d" % self.upscale_factor


class PixelUnshuffle(nn.Module):
    """reference ofa/utils.py:399-410; dtype-generic (the reference's one-hot kernel is fp32-only, Q7)."""

    def __init__(self, downscale_factor=2):
        super().__init__()
        self.downscale_factor = downscale_factor

    def forward(self, x):
        return ops.pixel_unshuffle(x, self.downscale_factor)

    def extra_repr(self):
        return "downscale_factor=%d" % self.downscale_factor


def pixel_unshuffle(input, downscale_factor):
    return ops.pixel_unshuffle(input, downscale_factor)


def build_pixelshuffle(upscale_factor=2):
    return PixelShuffle(upscale_factor)


def build_pixelunshuffle(downscale_factor=2):
    return PixelUnshuffle(downscale_factor)


_PLAIN_ACTS = {
    "relu": lambda inplace: nn.ReLU(inplace=inplace),
    "relu6": lambda inplace: nn.ReLU6(inplace=inplace),
    "tanh": lambda inplace: nn.Tanh(),
    "sigmoid": lambda inplace: nn.Sigmoid(),
    "prelu": lambda inplace: nn.PReLU(),
    "lrelu": lambda inplace: nn.LeakyReLU(0.1, inplace=inplace),
}


def build_activation(act_func, inplace=True, upscale_factor=2):
    """reference ofa/utils.py:242-306.  'pixelshuffle' alone always uses factor 2 there (:259-260);
    the '+act' variants honour `upscale_factor`.  h_swish / h_sigmoid are classification-only."""
    if act_func is None:
        return None
    if act_func in _PLAIN_ACTS:
        return _PLAIN_ACTS[act_func](inplace)
    if act_func == "pixelshuffle":
        return build_pixelshuffle(2)
    if act_func == "pixelunshuffle":
        return build_pixelunshuffle(2)
    if "+" in act_func:
        head, tail = act_func.split("+", 1)
        if head in ("pixelshuffle", "pixelunshuffle") and tail in _PLAIN_ACTS:
            first = build_pixelshuffle(upscale_factor) if head == "pixelshuffle" else build_pixelunshuffle(upscale_factor)
            return nn.Sequential(first, _PLAIN_ACTS[tail](inplace))
    raise ValueError("do not support: %s" % act_func)


# ------------------------------------------------------------------------------------ metric
def psnr(img1, img2):
    """reference ofa/utils.py:27-34 (uint8 images)."""
    assert img1.dtype == img2.dtype == np.uint8
    mse = np.mean((img1.astype(np.float64) - img2.astype(np.float64)) ** 2)
    if mse == 0:
        return float("inf")
    return 20 * math.log10(255.0 / math.sqrt(mse))


def _make_grid(t, nrow, padding=2):
    """the slice of torchvision.utils.make_grid the reference relies on (sr_run_manager.py:577):
    batch 1 -> squeeze; otherwise a zero-padded mosaic, `nrow` images per row."""
    if t.size(0) == 1:
        return t.squeeze(0)
    n, c, h, w = t.shape
    xmaps = min(nrow, n)
    ymaps = int(math.ceil(float(n) / xmaps))
    hh, ww = h + padding, w + padding
    grid = t.new_zeros((c, hh * ymaps + padding, ww * xmaps + padding))
    k = 0
    for yy in range(ymaps):
        for xx in range(xmaps):
            if k >= n:
                break
            grid[:, yy * hh + padding: yy * hh + padding + h, xx * ww + padding: xx * ww + padding + w] = t[k]
            k += 1
    return grid


def tensor2img_np(tensor, out_type=np.uint8, min_max=(0, 1)):
    """reference sr_run_manager.py:567-590 / progressive_shrinking.py:479-496.  Works on a COPY (the
    reference's in-place clamp only touches the caller's tensor on a CPU run)."""
    t = tensor.detach().float().cpu().clone().clamp_(*min_max)
    t = (t - min_max[0]) / (min_max[1] - min_max[0])
    if t.dim() == 4:
        img = _make_grid(t, nrow=int(math.sqrt(len(t)))).numpy()
        img = np.transpose(img, (1, 2, 0))
    elif t.dim() == 3:
        img = np.transpose(t.numpy(), (1, 2, 0))
    elif t.dim() == 2:
        img = t.numpy()
    else:
        raise TypeError("Only support 4D, 3D and 2D tensor. But received tensor with dimension = %d" % t.dim())
    if out_type == np.uint8:
        img = (img * 255.0).round()
    return img.astype(out_type)


def rgb2y(img):
    """BT.601 luma, rounded, uint8 (reference sr_run_manager.py:592-597)."""
    assert img.dtype == np.uint8
    return ((np.dot(img[..., :3], [65.481, 128.553, 24.966])) / 255.0 + 16.0).round().astype(np.uint8)


def psnr_y(output, target):
    """the reference's logged metric: psnr(rgb2y(tensor2img_np(out)), rgb2y(tensor2img_np(hr)))."""
    return psnr(rgb2y(tensor2img_np(output)), rgb2y(tensor2img_np(target)))


def psnr_y_per_image(output, target):
    """[psnr_y of every image of a batch on its own] -- what a batch-1 loader reports image by image.  The reference's
    batch PSNR is taken over a make_grid mosaic, so a size-bucketed batch (eval_ofa_net_sr.py) must be
    scored per image to reproduce the batch-1 numbers."""
    return [psnr_y(output[i:i + 1], target[i:i + 1]) for i in range(output.shape[0])]


def bucket_by_size(items, key=None, max_batch=None):
    """group tensors (or records: `key(item)` -> the [1,C,H,W] tensor) of EQUAL spatial size, keeping first-seen
    order: the size-bucketing BASELINE config 5 needs -- the reference evaluates Set14 at batch 1 because the images
    differ in size (div2k_setxx.py:182-190); zero-padding to a common size would change the
    pixels near the border (BN output of a padded zero is not zero), equal-size batching changes nothing.
    Returns a list of lists."""
    key = key or (lambda t: t)
    order, groups = [], {}
    for it in items:
        shp = tuple(key(it).shape[-2:])
        if shp not in groups:
            groups[shp] = [[]]
            order.append(shp)
        if max_batch and len(groups[shp][-1]) >= max_batch:
            groups[shp].append([])
        groups[shp][-1].append(it)
    return [chunk for shp in order for chunk in groups[shp]]


def device_batch(mini_batch, device):
    """a loader batch on `device` under the reference's keys.  Batches of a provider built with lr_on_device=True carry
    only 'image_u8' (uint8 HR): the LR images are then made on the GPU with PIL's exact bicubic arithmetic
    (ops.lr_images_from_u8) instead of by the host-side PIL calls of div2k_setxx.py:288-298."""
    if "image_u8" in mini_batch:
        return ops.lr_images_from_u8(mini_batch["image_u8"].to(device, non_blocking=True))
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in mini_batch.items()}


class AverageMeter(object):
    """reference ofa/utils.py:53-75"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


# ------------------------------------------------------------------------------- module bases
class MyModule(nn.Module):
    """reference ofa/utils.py:78-93"""

    # The inference operand cache (ops.py: BN-folded weight images kept across eval-mode forwards) is dropped whenever
    # weights may change behind autograd's version counters: back to training mode, a state dict loaded, a re-init.
    def train(self, mode=True):
        if mode:
            from . import ops
            ops.clear_infer_cache()
        return super().train(mode)

    def load_state_dict(self, *args, **kwargs):
        from . import ops
        ops.clear_infer_cache()
        return super().load_state_dict(*args, **kwargs)

    def forward(self, x):
        raise NotImplementedError

    @property
    def module_str(self):
        raise NotImplementedError

    @property
    def config(self):
        raise NotImplementedError

    @staticmethod
    def build_from_config(config):
        raise NotImplementedError


class MyNetwork(MyModule):
    """reference ofa/utils.py:96-186"""

    def zero_last_gamma(self):
        raise NotImplementedError

    def set_bn_param(self, momentum, eps):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.momentum = momentum
                m.eps = eps

    def get_bn_param(self):
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                return {"momentum": m.momentum, "eps": m.eps}
        return None

    def init_model(self, model_init):
        """he_fout / he_fin for convs, BN gamma=1 beta=0 (reference ofa/utils.py:134-155).  Transform
        matrices are not nn.Conv2d and keep their identity init."""
        from . import ops
        ops.clear_infer_cache()
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                kk = m.kernel_size[0] * m.kernel_size[1]
                if model_init == "he_fout":
                    fan = kk * m.out_channels
                elif model_init == "he_fin":
                    fan = kk * m.in_channels
                else:
                    raise NotImplementedError
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                stdv = 1.0 / math.sqrt(m.weight.size(1))
                m.weight.data.uniform_(-stdv, stdv)
                if m.bias is not None:
                    m.bias.data.zero_()

    def get_parameters(self, keys=None, mode="include", exclude_set=None):
        """name-substring parameter filter used for the weight-decay groups (reference :157-183)."""
        exclude_set = exclude_set or {}
        if mode not in ("include", "exclude"):
            raise ValueError("do not support: %s" % mode)
        for name, param in self.named_parameters():
            if name in exclude_set:
                continue
            if keys is None:
                yield param
                continue
            hit = any(key in name for key in keys)
            if hit == (mode == "include"):
                yield param

    def weight_parameters(self, exclude_set=None):
        return self.get_parameters(exclude_set=exclude_set)


def psnr_y_device(output, target):
    """psnr_y evaluated ON THE DEVICE, bit-for-bit the same quantisation steps (clamp, *255, round-half-even,
    BT.601 luma in fp64, round) and, for batches > 1, the same pixel count as the reference's
    make_grid mosaic (2-px zero padding contributes zero error but is counted).
    Returns a 0-dim fp64 tensor: no host sync until the caller reads it."""
    def luma(t):
        u8 = (t.detach().float().clamp(0, 1) * 255.0).round().double()
        y = (u8[:, 0] * 65.481 + u8[:, 1] * 128.553 + u8[:, 2] * 24.966) / 255.0 + 16.0
        return y.round()

    n, _, h, w = output.shape
    sq = ((luma(output) - luma(target)) ** 2).sum()
    if n == 1:
        count = h * w
    else:
        xmaps = min(int(math.sqrt(n)), n)
        ymaps = int(math.ceil(float(n) / xmaps))
        count = ((h + 2) * ymaps + 2) * ((w + 2) * xmaps + 2)
    mse = sq / count
    return 20.0 * torch.log10(255.0 / torch.sqrt(mse))
