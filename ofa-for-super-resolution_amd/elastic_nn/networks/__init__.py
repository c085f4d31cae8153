from .ofa_mbs4 import OFAMobileNetS4  # noqa: F401
