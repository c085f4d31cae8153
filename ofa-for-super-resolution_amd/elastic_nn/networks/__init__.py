from .ofa_mbs4 import OFAMobileNetS4  # noqa: F401
from .ofa_mbx4 import OFAMobileNetX4  # noqa: F401
