from .proxyless_nets import MobileInvertedResidualBlock  # noqa: F401
from .mobilenet_s4 import MobileNetS4  # noqa: F401
