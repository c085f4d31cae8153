"""import-path mirror of the reference's ofa/imagenet_codebase/utils (SR-relevant names only)."""
from ...utils import (AverageMeter, MyModule, MyNetwork, build_activation, get_net_device, get_same_padding,  # noqa: F401
                      int2list, list_mean, make_divisible, sub_filter_start_end, subset_mean)
from .pytorch_utils import count_net_flops, count_parameters, get_net_info  # noqa: F401
