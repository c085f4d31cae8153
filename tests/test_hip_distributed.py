"""N > 1 path on the REAL S4 supernet with the HIP kernels: two ranks (gloo process group, both on GPU 0 -- the box
has one card; on a node the same code runs with backend 'nccl' = RCCL, one rank per GPU).  `-m gpu`.

What is exercised is the interaction the toy-model CPU test (tests/test_distributed_gloo.py) cannot reach: the
composite MB blocks' DEFERRED weight gradients (ops._flush_deferred at the end of backward) feeding
FlatGradReducer(gather=True) -- the mode the trainers and bench.py use -- through sub-network sampling with the shared
seed rule and gradient accumulation over two sub-steps.  Checked on every rank: the reduced gradient of every parameter
equals the mean of the two ranks' single-process shard gradients; the grad-None set of the sampled sub-networks is
preserved (Adam skips those parameters, reference semantics); weights are identical on both ranks after the Adam step;
BN statistics stay rank-local (reference: nn.DataParallel replicas, sr_run_manager.py:197-198)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu


def _worker(rank, world, init_file, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    import random
    import torch.nn.functional as F
    dd = importlib.import_module(PKG + ".distributed")
    dop = importlib.import_module(PKG + ".elastic_nn.modules.dynamic_op")
    nets = importlib.import_module(PKG + ".elastic_nn.networks")
    ops = importlib.import_module(PKG + ".ops")
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        assert ops.DEFER_WGRAD, "the test is about the deferred weight gradients"
        dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
        torch.manual_seed(50 + rank)                 # different init per rank ...
        net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                                  pixelshuffle_depth_list=[2])
        net.init_model("he_fout")
        net.to(dev).train()
        dd.broadcast_module(net)                     # ... made identical here
        sd0 = {k: v.detach().clone() for k, v in net.state_dict().items()}
        g = torch.Generator().manual_seed(3)
        hr = torch.rand((world, 2, 3, 64, 64), generator=g)
        lr = F.interpolate(hr.flatten(0, 1), scale_factor=0.25, mode="bicubic", antialias=True).clamp_(0, 1)
        lr = lr.view(world, 2, 3, 16, 16).to(dev)
        hr = hr.to(dev)
        seeds = [int("%d%.3d%.3d" % (7, sub, 0)) for sub in range(2)]    # progressive_shrinking.py:164

        def backward_on(shard):
            for s in seeds:                          # dynamic_batch_size = 2: gradients accumulate over sub-steps
                random.seed(s)
                net.sample_active_subnet()
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    out = net(lr[shard])
                F.mse_loss(out.float(), hr[shard]).backward()
                ops.flush_deferred()          # deferred weight gradients -> .grad (ops.py)

        # single-process shard gradients (what each rank would compute alone), from the same starting state
        ref = []
        for shard in range(world):
            net.load_state_dict(sd0)
            net.zero_grad(set_to_none=True)
            backward_on(shard)
            torch.cuda.synchronize()
            ref.append([None if p.grad is None else p.grad.detach().clone() for p in net.parameters()])
        none_set = [gr is None for gr in ref[0]]
        assert none_set == [gr is None for gr in ref[1]], "ranks share the sub-network => the same untouched set"
        assert any(none_set) and not all(none_set)

        net.load_state_dict(sd0)
        net.zero_grad(set_to_none=True)
        params = list(net.parameters())
        reducer = dd.FlatGradReducer(params, gather=True)
        assert reducer.nbytes == 4 * sum(p.numel() for p in params)
        opt = torch.optim.Adam(params, lr=1e-3, weight_decay=3e-5)
        reducer.prepare()
        backward_on(rank)
        reducer.reduce()
        for i, p in enumerate(params):
            assert (p.grad is None) == none_set[i], "grad-None set changed by the exchange (parameter %d)" % i
            if p.grad is not None:
                want = 0.5 * (ref[0][i] + ref[1][i])
                scale = float(want.abs().max()) + 1e-12
                assert float((p.grad - want).abs().max()) <= 1e-5 * scale + 1e-9, "reduced gradient != mean of the shards"
                assert p.grad.data_ptr() >= reducer.flat.data_ptr()      # a view of the flat bucket
        opt.step()
        flat = torch.cat([p.detach().reshape(-1) for p in params]).cpu()
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered), "weights diverged across ranks after the step"
        changed = [not torch.equal(p.detach(), sd0[n]) for (n, _), p in zip(net.named_parameters(), params)]
        assert changed == [not u for u in none_set], "Adam must skip exactly the untouched parameters"
        # BN statistics are rank-local (different shards => different running means)
        rm = net.dec_first_conv_block.bn.running_mean.detach().cpu()
        both = [torch.zeros_like(rm) for _ in range(world)]
        dist.all_gather(both, rm)
        assert not torch.equal(both[0], both[1])
        single = [None if p.grad is None else p.grad.detach().clone() for p in params]
        reducer.remove()

        # the two-bucket overlapped exchange (early_params = the decoder tail, started from the backward milestone while
        # the MB stack's backward still runs) gives the single bucket's result bit for bit
        net.load_state_dict(sd0)
        net.zero_grad(set_to_none=True)
        red2 = dd.FlatGradReducer(params, gather=True, early_params=net.early_gradient_parameters())
        assert 0 < red2.n_early < len(red2.params) and red2.nbytes == reducer.nbytes
        fired = []
        orig = red2._on_tail
        red2._on_tail = lambda: (fired.append(red2._armed), orig())[1]
        red2._removers.append(ops.register_grad_milestone("decoder_tail", red2._on_tail))
        red2.prepare()
        for si, s in enumerate(seeds):
            random.seed(s)
            net.sample_active_subnet()
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = net(lr[rank])
            if si == len(seeds) - 1:
                red2.arm()                           # only the last pass of the step may start the exchange
            F.mse_loss(out.float(), hr[rank]).backward()
            ops.flush_deferred()
        assert fired == [False, True], fired
        assert red2._early is not None, "the early bucket's exchange was not started from the backward pass"
        red2.reduce()
        for i, p in enumerate(params):
            assert (p.grad is None) == (single[i] is None), i
            if p.grad is not None:
                assert torch.equal(p.grad, single[i]), "two-bucket exchange != single bucket (parameter %d)" % i
        red2.remove()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_s4_two_rank_step_with_deferred_weight_grads():
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, os.path.join(d, "rendezvous"), d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok0")) and os.path.exists(os.path.join(d, "ok1"))
