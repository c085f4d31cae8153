"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/ofasr.h declares; the host ops refuse to run without the GPU (no silent fallback)."""
import os
import re

import pytest
import torch

from conftest import ROOT, amd


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ofasr.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofasr_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    C = amd("_C")
    if not os.path.exists(C.LIB_PATH):
        C.build()
    L = C.lib()
    declared = _declared_symbols()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), "libofasr_hip.so does not export %s" % name
    assert set(declared) == set(C.SIGNATURES), "ctypes table and header drifted apart"
    hdr = open(os.path.join(ROOT, "include", "ofasr.h")).read()
    assert L.ofasr_version() == int(re.search(r"#define OFASR_VERSION (\d+)", hdr).group(1)) >= 300
    assert L.ofasr_status_string(-2) == b"unsupported shape or dtype"


def test_workspace_queries_are_host_only():
    L = amd("_C").lib()
    # N=16, C=384, 64x64, k=7: one partial per (part, c, tap)
    n = L.ofasr_dwconv_wgrad_workspace(16, 384, 64, 64, 7)
    assert n > 0 and n % (384 * 49 * 4) == 0
    assert L.ofasr_pwconv_wgrad_workspace(16, 64, 384, 4096) > 0
    assert L.ofasr_pwconv_wgrad_workspace(0, 64, 384, 4096) == 0


def test_argument_validation_without_gpu():
    # validation happens before any launch, so it can be exercised on a GPU-less host
    import ctypes
    C = amd("_C")
    L = C.lib()
    rc = L.ofasr_pixel_shuffle(None, None, 1, 1, 1, 1, 2, 4, None)
    assert rc == -1 and b"null" in L.ofasr_last_error_string()
    buf = ctypes.create_string_buffer(64)
    p = ctypes.cast(buf, ctypes.c_void_p)
    assert L.ofasr_dwconv_fwd(p, p, p, 1, 1, 4, 4, 4, 0, None) == -2      # even kernel size
    assert L.ofasr_pwconv_fwd(p, p, 2, p, 1, 4, 4, 4, 0, None) == -1      # ldw < Cin
    with pytest.raises(C.OfasrError):
        C.check(-2, "probe")


def test_ops_refuse_cpu_tensors():
    ops = amd("ops")
    C = amd("_C")
    x = torch.zeros(1, 4, 2, 2)
    with pytest.raises(C.OfasrError):
        ops.pixel_shuffle(x, 2)
    with pytest.raises(C.OfasrError):
        ops.pwconv(torch.zeros(1, 4, 2, 2), torch.zeros(8, 4, 1, 1), 8)
    with pytest.raises(C.OfasrError):
        ops.dwconv(torch.zeros(1, 4, 2, 2), torch.zeros(4, 1, 3, 3))
