"""MI355X-native OFA-SR hot path.

The directory name carries hyphens, so import it through importlib:

    import importlib
    ofa_amd = importlib.import_module("ofa-for-super-resolution_amd")

Sub-modules mirror the reference's layout (`elastic_nn.modules.dynamic_op`, ...); every hot op
runs in the HIP library `csrc/libofasr_hip.so` through the C ABI in `include/ofasr.h`.  There is
no CPU fallback: ops raise if the library is missing or the tensors are not on an AMD GPU.
"""
__version__ = "0.1.0"
