from .dynamic_op import DynamicSeparableConv2d, DynamicPointConv2d, DynamicBatchNorm2d
from .dynamic_layers import DynamicMBConvLayer
