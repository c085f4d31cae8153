"""Deferred weight gradients of the composite MB block (ops.py `_Deferred`, include/ofasr.h ofasr_mbconv_defer_join /
ofasr_mbconv_join): the weight-gradient kernels run on the library's side stream and are joined once at the end of the
backward pass instead of once per block.  Results must be bit-identical to the immediate mode, including gradient
accumulation over several backward passes (dynamic_batch_size > 1, reference progressive_shrinking.py:152-199),
the package's deferred-gradient hooks (the data-parallel bucket relies on them) and a scratch buffer shared by all blocks."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import amd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    dop = amd("elastic_nn.modules.dynamic_op")
    nets = amd("elastic_nn.networks")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    torch.manual_seed(0)
    return nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                               pixelshuffle_depth_list=[1, 2]).to(DEV).train()


def _two_pass_grads(net, lr, hr, hooks=None):
    net.zero_grad(set_to_none=True)
    for sub in range(2):
        random.seed(100 + sub)
        net.sample_active_subnet()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(lr)
        F.mse_loss(out.float(), hr).backward()
        amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    torch.cuda.synchronize()
    return {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in net.named_parameters()}


@pytest.mark.parametrize("shared_tmp", [False, True])
def test_deferred_equals_immediate_with_accumulation(shared_tmp):
    ops = amd("ops")
    net = _net()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    lr = torch.rand(4, 3, 32, 32, device=DEV)
    hr = torch.rand(4, 3, 128, 128, device=DEV)
    was, was_tmp = ops.deferred_weight_grads(False), ops.SHARED_TMP
    try:
        ref = _two_pass_grads(net, lr, hr)
        net.load_state_dict(state)
        ops.deferred_weight_grads(True)
        ops.SHARED_TMP = shared_tmp
        got = _two_pass_grads(net, lr, hr)
        assert not ops._Deferred.keep and not ops._Deferred.grads and not ops._Deferred.queued
    finally:
        ops.deferred_weight_grads(was)
        ops.SHARED_TMP = was_tmp
    assert set(ref) == set(got)
    n_def = 0
    for n in ref:
        assert (ref[n] is None) == (got[n] is None), "None-ness of %s.grad differs" % n
        if ref[n] is not None:
            assert torch.equal(ref[n], got[n]), n
            n_def += 1
    assert n_def > 50


def test_deferred_grad_hooks_and_safety_nets():
    """gradients that flush_deferred() accumulates run the hooks of ops.register_deferred_grad_hook (the public stand-in
    for torch's post-accumulate hooks, which FlatGradReducer uses); without an explicit flush the optimizer-step pre-hook
    and the next composite forward settle a pending backward pass (all public torch API; the engine callback is opt-in)."""
    ops = amd("ops")
    assert not ops.DEFER_ENGINE_CALLBACK, "the default mode must not depend on torch's private engine callback"
    net = _net()
    seen = []
    w = net.blocks[0].mobile_inverted_conv.point_linear.conv.conv.weight
    dw = net.blocks[0].mobile_inverted_conv.depth_conv.conv.conv.weight
    rm = [ops.register_deferred_grad_hook(w, lambda p: seen.append(("pl", p.grad is not None))),
          ops.register_deferred_grad_hook(dw, lambda p: seen.append(("dw", p.grad is not None)))]
    was = ops.deferred_weight_grads(True)
    try:
        random.seed(3)
        net.sample_active_subnet()
        x = torch.rand(2, 3, 32, 32, device=DEV)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
        out.float().square().mean().backward()
        assert w.grad is None and ops._Deferred.grads, "deferred mode: the weight gradients are pending after backward()"
        ops.flush_deferred()
        assert sorted(seen) == [("dw", True), ("pl", True)] and w.grad is not None and dw.grad is not None
        ref = w.grad.clone()
        # (1) the optimizer-step pre-hook
        net.zero_grad(set_to_none=True)
        opt = torch.optim.SGD([w], lr=0.0)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
        out.float().square().mean().backward()
        assert w.grad is None
        opt.step()
        assert w.grad is not None and torch.equal(w.grad, ref) and not ops._Deferred.grads
        # (2) the next composite forward
        net.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(x)
        out.float().square().mean().backward()
        assert w.grad is None
        with torch.autocast("cuda", dtype=torch.bfloat16):
            net(x)
        assert w.grad is not None and torch.equal(w.grad, ref)
    finally:
        ops.deferred_weight_grads(was)
        for r in rm:
            r()


def test_autograd_grad_falls_back_outside_accumulate_callers():
    """with the mode off, torch.autograd.grad returns the composite's weight gradients as usual."""
    ops = amd("ops")
    net = _net()
    w = net.blocks[0].mobile_inverted_conv.point_linear.conv.conv.weight
    was = ops.deferred_weight_grads(False)
    try:
        random.seed(3)
        net.sample_active_subnet()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            out = net(torch.rand(2, 3, 32, 32, device=DEV))
        (gw,) = torch.autograd.grad(out.float().square().mean(), [w])
    finally:
        ops.deferred_weight_grads(was)
    assert gw is not None and float(gw.abs().sum()) > 0 and w.grad is None


def test_static_conv_weight_gradients_on_the_side_stream():
    """the optional deferral of the static convs' weight gradients (ops.CONV_DEFER_WGRAD) gives identical gradients."""
    ops = amd("ops")
    net = _net()
    lr = torch.rand(2, 3, 32, 32, device=DEV)
    hr = torch.rand(2, 3, 128, 128, device=DEV)
    was, was_conv = ops.deferred_weight_grads(True), ops.CONV_DEFER_WGRAD
    try:
        ops.CONV_DEFER_WGRAD = False
        ref = _two_pass_grads(net, lr, hr)
        ops.CONV_DEFER_WGRAD = True
        got = _two_pass_grads(net, lr, hr)
    finally:
        ops.deferred_weight_grads(was)
        ops.CONV_DEFER_WGRAD = was_conv
    for n in ref:
        assert (ref[n] is None) == (got[n] is None), n
        if ref[n] is not None:
            assert torch.equal(ref[n], got[n]), n


@pytest.mark.parametrize("dtype", [torch.bfloat16, None])
def test_mb_stack_call_equals_block_by_block(dtype):
    """ops.FusedMBStackFn (one autograd node / foreign call for all active MB blocks, ofasr_mbstack_fwd / _bwd) against
    the block-by-block path (ops.FusedMBConvFn per block): same kernels in the same order, so outputs, every gradient and
    its None-ness, and the BN buffers are bit-identical; sampled sub-networks with skipped blocks, bf16 and fp32."""
    ops, C = amd("ops"), amd("_C")
    net = _net()
    state = {k: v.clone() for k, v in net.state_dict().items()}
    lr = torch.rand(2, 3, 32, 32, device=DEV)
    hr = torch.rand(2, 3, 128, 128, device=DEV)

    def run(stack):
        was = ops.FUSED_STACK
        ops.FUSED_STACK = stack
        try:
            net.load_state_dict(state)
            net.zero_grad(set_to_none=True)
            outs = []
            for sub in range(2):                       # two sub-steps: gradients accumulate
                random.seed(200 + sub)
                net.sample_active_subnet()
                C.reset_launch_counts()
                if dtype is None:
                    out = net(lr)
                else:
                    with torch.autocast("cuda", dtype=dtype):
                        out = net(lr)
                F.mse_loss(out.float(), hr).backward()
                ops.flush_deferred()
                outs.append(out.detach().float().clone())
            torch.cuda.synchronize()
            grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in net.named_parameters()}
            bufs = {k: v.clone() for k, v in net.state_dict().items() if "running_" in k or "num_batches" in k}
            return outs, grads, bufs
        finally:
            ops.FUSED_STACK = was

    o1, g1, b1 = run(True)
    o0, g0, b0 = run(False)
    for a, b in zip(o1, o0):
        assert torch.equal(a, b)
    for n in g0:
        assert (g0[n] is None) == (g1[n] is None), n
        if g0[n] is not None:
            assert torch.equal(g0[n], g1[n]), n
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k
    assert sum(g is None for g in g0.values()) > 0 and sum(g is not None for g in g0.values()) > 100
