#!/usr/bin/env python3
"""Per-kernel micro-benchmark of the C-ABI entry points at the bench shapes (N=16, 64x64, 64<->384 channels).
Times each launch with events on the launch stream; prints achieved algorithmic GB/s.
usage: python tools/kbench.py [--lib path/to/lib.so] [--dtype bf16|f32] [--reps 50] [--only substr]"""
import argparse
import ctypes
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "ofa-for-super-resolution_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--only", default="")
    ap.add_argument("--N", type=int, default=16)
    ap.add_argument("--S", type=int, default=64)
    ap.add_argument("--mid", type=int, default=384, help="expanded width (192 / 256 / 384 for e = 3 / 4 / 6)")
    a = ap.parse_args()
    import torch
    C = importlib.import_module(PKG + "._C")
    if a.lib:
        C.LIB_PATH = os.path.abspath(a.lib)
    L = C.lib()
    dt = {"bf16": torch.bfloat16, "f32": torch.float32, "f16": torch.float16}[a.dtype]
    code = {"bf16": 2, "f32": 0, "f16": 1}[a.dtype]
    es = 2 if a.dtype != "f32" else 4
    dev = "cuda:0"
    N, S, mid = a.N, a.S, a.mid
    HW = S * S
    x64 = torch.randn(N, 64, S, S, device=dev).to(dt)
    xm = torch.randn(N, mid, S, S, device=dev).to(dt)
    ym = torch.empty_like(xm)
    y64 = torch.empty_like(x64)
    w1 = torch.randn(mid, 64, 1, 1, device=dev) * 0.1
    w2 = torch.randn(64, mid, 1, 1, device=dev) * 0.1
    dw1 = torch.zeros_like(w1)
    dw2 = torch.zeros_like(w2)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
    wsn = ctypes.c_size_t(ws.numel())
    cases = []
    act_big = (N * (64 + mid) * HW) * es
    cases.append(("pwconv_fwd 64->mid", act_big + 4 * 64 * mid,
                  lambda: L.ofasr_pwconv_fwd(P(x64), P(w1), 64, P(ym), N, 64, mid, HW, code, st)))
    cases.append(("pwconv_fwd mid->64", act_big + 4 * 64 * mid,
                  lambda: L.ofasr_pwconv_fwd(P(xm), P(w2), mid, P(y64), N, mid, 64, HW, code, st)))
    cases.append(("pwconv_dgrad mid->64 (of expand)", act_big,
                  lambda: L.ofasr_pwconv_dgrad(P(xm), P(w1), 64, P(y64), N, 64, mid, HW, code, st)))
    cases.append(("pwconv_dgrad 64->mid (of project)", act_big,
                  lambda: L.ofasr_pwconv_dgrad(P(x64), P(w2), mid, P(ym), N, mid, 64, HW, code, st)))
    cases.append(("pwconv_wgrad expand", act_big,
                  lambda: L.ofasr_pwconv_wgrad(P(xm), P(x64), P(dw1), 64, N, 64, mid, HW, code, P(ws), wsn, st)))
    cases.append(("pwconv_wgrad project", act_big,
                  lambda: L.ofasr_pwconv_wgrad(P(x64), P(xm), P(dw2), mid, N, mid, 64, HW, code, P(ws), wsn, st)))
    for K in (3, 5, 7):
        f = torch.randn(mid, 1, K, K, device=dev) * 0.1
        df = torch.zeros_like(f)
        b = 2 * N * mid * HW * es
        cases.append(("dwconv_fwd k%d" % K, b, lambda f=f, K=K: L.ofasr_dwconv_fwd(P(xm), P(f), P(ym), N, mid, S, S, K, code, st)))
        cases.append(("dwconv_wgrad k%d" % K, b, lambda df=df, K=K: L.ofasr_dwconv_wgrad(P(ym), P(xm), P(df), N, mid, S, S, K, code, P(ws), wsn, st)))
    sc = torch.ones(4, mid, device=dev)
    cases.append(("bn_stats mid", N * mid * HW * es, lambda: L.ofasr_bn_stats(P(xm), N, mid, HW, code, P(ws), wsn, st)))
    cases.append(("bn_act_fwd mid relu6", 2 * N * mid * HW * es,
                  lambda: L.ofasr_bn_act_fwd(P(xm), None, P(ym), P(sc[0]), P(sc[1]), P(sc[2]), N, mid, HW, 1, code, st)))
    dg = torch.zeros(mid, device=dev)
    cases.append(("bn_act_bwd mid relu6", 5 * N * mid * HW * es,
                  lambda: L.ofasr_bn_act_bwd(P(ym), P(xm), None, P(ym), None, P(sc[0]), P(sc[1]), P(sc[2]), P(sc[3]),
                                             P(dg), P(dg), N, mid, HW, 1, 1, code, P(ws), wsn, st)))
    x256 = torch.randn(N, 256, S, S, device=dev).to(dt)
    y256 = torch.empty(N, 64, 2 * S, 2 * S, device=dev, dtype=dt)
    cases.append(("pixel_shuffle 256ch", 2 * x256.numel() * es,
                  lambda: L.ofasr_pixel_shuffle(P(x256), P(y256), N, 64, S, S, 2, es, st)))
    import torch.nn.functional as F
    flops = {}
    for (ci, co, hh, K) in ((3, 64, S, 5), (64, 256, S, 5), (64, 256, 2 * S, 5), (64, 3, 4 * S, 5), (64, 64, S, 3)):
        if a.dtype == "f32":
            break
        xc = torch.randn(N, ci, hh, hh, device=dev).to(dt)
        yc = torch.randn(N, co, hh, hh, device=dev).to(dt)
        wc = torch.randn(co, ci, K, K, device=dev) * 0.05
        w16 = wc.to(dt)
        nb = (xc.numel() + yc.numel()) * es
        fl = 2.0 * N * hh * hh * ci * co * K * K
        for nm, fn in (("conv2d_fwd %d->%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K: L.ofasr_conv2d_fwd(
                            P(xc), P(wc), P(yc), N, ci, co, hh, hh, K, code, P(ws), wsn, st)),
                       ("conv2d_dgrad %d<-%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K: L.ofasr_conv2d_dgrad(
                            P(yc), P(wc), P(xc), N, ci, co, hh, hh, K, code, P(ws), wsn, st)),
                       ("conv2d_wgrad %d->%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K: L.ofasr_conv2d_wgrad(
                            P(yc), P(xc), P(wc), N, ci, co, hh, hh, K, code, P(ws), wsn, st)),
                       ("miopen  fwd %d->%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, w16=w16, K=K: (F.conv2d(xc, w16, padding=K // 2), 0)[1]),
                       ("miopen  bwd(dx,dw) %d->%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, yc=yc, w16=w16, K=K: (torch.ops.aten.convolution_backward(
                            yc, xc, w16, None, [1, 1], [K // 2, K // 2], [1, 1], False, [0, 0], 1, [True, True, False]), 0)[1]),
                       ("miopen  bwd(dw) %d->%d k%d @%d" % (ci, co, K, hh),
                        lambda xc=xc, yc=yc, w16=w16, K=K: (torch.ops.aten.convolution_backward(
                            yc, xc, w16, None, [1, 1], [K // 2, K // 2], [1, 1], False, [0, 0], 1, [False, True, False]), 0)[1])):
            cases.append((nm, nb, fn))
            flops[nm] = fl
    if a.dtype == "f32":   # fp32 static convs: own kernels on v_mfma_f32_32x32x2_f32 against the vendor's
        for (ci, co, hh, K) in ((3, 64, S, 5), (64, 256, S, 5), (64, 256, 2 * S, 5), (64, 3, 4 * S, 5)):
            xc = torch.randn(N, ci, hh, hh, device=dev)
            yc = torch.randn(N, co, hh, hh, device=dev)
            wc = torch.randn(co, ci, K, K, device=dev) * 0.05
            nb = (xc.numel() + yc.numel()) * 4
            fl = 2.0 * N * hh * hh * ci * co * K * K
            wsf = torch.empty(max(L.ofasr_conv2d_f32_workspace(ci, co, K, 0), L.ofasr_conv2d_f32_workspace(ci, co, K, 1),
                                  L.ofasr_conv2d_f32_wgrad_workspace(N, ci, co, hh, hh, K), 16), dtype=torch.uint8, device=dev)
            wn = wsf.numel()
            for nm, fn in (("conv2d_f32_fwd %d->%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K, wsf=wsf, wn=wn: L.ofasr_conv2d_f32_fwd(
                                P(xc), P(wc), P(yc), N, ci, co, hh, hh, K, P(wsf), wn, st)),
                           ("conv2d_f32_dgrad %d<-%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K, wsf=wsf, wn=wn: L.ofasr_conv2d_f32_dgrad(
                                P(yc), P(wc), P(xc), N, ci, co, hh, hh, K, P(wsf), wn, st)),
                           ("conv2d_f32_wgrad %d->%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, yc=yc, wc=wc, ci=ci, co=co, hh=hh, K=K, wsf=wsf, wn=wn: L.ofasr_conv2d_f32_wgrad(
                                P(yc), P(xc), P(wc), N, ci, co, hh, hh, K, P(wsf), wn, st)),
                           ("miopen32 fwd %d->%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, wc=wc, K=K: (F.conv2d(xc, wc, padding=K // 2), 0)[1]),
                           ("miopen32 bwd(dx) %d->%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, yc=yc, wc=wc, K=K: (torch.ops.aten.convolution_backward(
                                yc, xc, wc, None, [1, 1], [K // 2, K // 2], [1, 1], False, [0, 0], 1, [True, False, False]), 0)[1]),
                           ("miopen32 bwd(dw) %d->%d k%d @%d" % (ci, co, K, hh),
                            lambda xc=xc, yc=yc, wc=wc, K=K: (torch.ops.aten.convolution_backward(
                                yc, xc, wc, None, [1, 1], [K // 2, K // 2], [1, 1], False, [0, 0], 1, [False, True, False]), 0)[1])):
                cases.append((nm, nb, fn))
                flops[nm] = fl
    torch.backends.cudnn.benchmark = True
    for name, nbytes, fn in cases:
        if a.only and a.only not in name:
            continue
        for _ in range(5):
            rc = fn()
            assert rc == 0, (name, rc, L.ofasr_last_error_string())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / a.reps
        extra = "  %7.1f TFLOP/s" % (flops[name] / us / 1e6) if name in flops else ""
        print("%-36s %8.1f us  %8.1f GB/s  (%.1f MB algorithmic)%s" % (name, us, nbytes / us / 1e3, nbytes / 1e6, extra))


if __name__ == "__main__":
    main()
