"""N > 1 path on CPU: two processes, `gloo` backend (the same code runs on RCCL with backend 'nccl').
Covers distributed.FlatGradReducer (single flat all-reduce, grad-None preservation for parameters the
sampled sub-network does not touch, gradient accumulation over sub-steps) and broadcast_module.
The modules used here are plain torch layers: the HIP ops need a GPU, the exchange logic does not."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

from conftest import ROOT, PKG


class _Elastic(nn.Module):
    """a toy 'supernet': branch b is skipped when depth == 1, like elastic depth."""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 6)
        self.b = nn.Linear(6, 6)
        self.c = nn.Linear(6, 2)
        self.depth = 2

    def forward(self, x):
        x = torch.tanh(self.a(x))
        if self.depth > 1:
            x = torch.tanh(self.b(x))
        return self.c(x)


def _worker(rank, world, init_file, out_dir, gather=False):
    sys.path.insert(0, ROOT)
    import importlib
    dd = importlib.import_module(PKG + ".distributed")
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        torch.manual_seed(100 + rank)           # different init per rank ...
        net = _Elastic()
        dd.broadcast_module(net)                # ... made identical here
        ref = _Elastic()
        torch.manual_seed(100)
        ref = _Elastic()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.equal(p, q), "broadcast_module must copy rank 0's parameters"

        reducer = dd.FlatGradReducer(net.parameters(), gather=gather)
        assert reducer.nbytes == 4 * sum(p.numel() for p in net.parameters())
        opt = torch.optim.Adam(net.parameters(), lr=1e-2, weight_decay=1e-2)
        g = torch.Generator().manual_seed(7)
        data = torch.randn(2, 2, 8, 6, generator=g)     # [step][rank][batch][feat]
        tgt = torch.randn(2, 2, 8, 2, generator=g)

        # single-process oracle: same model, the concatenated global batch
        oracle = _Elastic()
        oracle.load_state_dict(ref.state_dict())
        oopt = torch.optim.Adam(oracle.parameters(), lr=1e-2, weight_decay=1e-2)

        for step, depth in enumerate([1, 2]):
            net.depth = oracle.depth = depth
            reducer.prepare()
            # two accumulation sub-steps (dynamic_batch_size = 2) on the same data
            for _ in range(2):
                nn.functional.mse_loss(net(data[step, rank]), tgt[step, rank]).backward()
            reducer.reduce()
            untouched = [p.grad is None for p in net.parameters()]
            if depth == 1:
                assert untouched == [False, False, True, True, False, False], untouched
            else:
                assert not any(untouched)
            opt.step()

            oopt.zero_grad(set_to_none=True)
            for _ in range(2):
                nn.functional.mse_loss(oracle(data[step].reshape(16, 6)), tgt[step].reshape(16, 2)).backward()
            oopt.step()
            for p, q in zip(net.parameters(), oracle.parameters()):
                assert torch.allclose(p, q, rtol=1e-5, atol=1e-6), "DP step != single-process step on the global batch"
        # every rank ends with identical weights
        flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        assert all(torch.equal(gathered[0], t) for t in gathered)
        reducer.remove()
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("gather", [False, True])
def test_flat_grad_reducer_two_ranks_gloo(gather):
    """gather=False: .grad are bucket views during backward; gather=True (what the GPU trainers use): gradients are
    copied into the bucket by one multi-tensor copy in reduce().  Same DP step either way."""
    with tempfile.TemporaryDirectory() as d:
        init_file = os.path.join(d, "rendezvous")
        mp.spawn(_worker, args=(2, init_file, d, gather), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok0")) and os.path.exists(os.path.join(d, "ok1"))


def test_reducer_single_process_semantics():
    """world size 1 (no process group): prepare/reduce still give grad-None for untouched parameters."""
    import importlib
    dd = importlib.import_module(PKG + ".distributed")
    net = _Elastic()
    red = dd.FlatGradReducer(net.parameters())
    net.depth = 1
    red.prepare()
    assert all(p.grad is not None for p in net.parameters())
    xin = torch.randn(3, 6)
    net(xin).sum().backward()
    red.reduce()
    assert [p.grad is None for p in net.parameters()] == [False, False, True, True, False, False]
    # gradients are views of one flat buffer
    assert net.a.weight.grad.data_ptr() == red.flat.data_ptr()
    # gather mode: no .grad during backward, the same views and None-ness after reduce()
    want = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    red.remove()
    red2 = dd.FlatGradReducer(net.parameters(), gather=True)
    red2.prepare()
    assert all(p.grad is None for p in net.parameters())
    net(xin).sum().backward()
    red2.reduce()
    for p, w in zip(net.parameters(), want):
        assert (p.grad is None) == (w is None)
        if w is not None:
            assert torch.equal(p.grad, w)
    assert net.a.weight.grad.data_ptr() == red2.flat.data_ptr()
    assert dd.world_size() == 1 and dd.rank() == 0 and not dd.is_distributed()


class _Tail(nn.Module):
    """toy net with the backward milestone of the S4 net: head / mid are the "decoder tail" (their backward runs first)"""

    def __init__(self, ops):
        super().__init__()
        self.ops = ops
        self.stem = nn.Linear(6, 6)
        self.body = nn.Linear(6, 6)
        self.mid = nn.Linear(6, 6)
        self.head = nn.Linear(6, 2)
        self.depth = 2

    def forward(self, x):
        x = torch.tanh(self.stem(x))
        if self.depth > 1:
            x = torch.tanh(self.body(x))
        x = self.ops.grad_milestone(x, "decoder_tail")
        return self.head(torch.tanh(self.mid(x)))

    def early_gradient_parameters(self):
        return list(self.mid.parameters()) + list(self.head.parameters())


def _worker_overlap(rank, world, init_file, out_dir):
    sys.path.insert(0, ROOT)
    import importlib
    dd = importlib.import_module(PKG + ".distributed")
    ops = importlib.import_module(PKG + ".ops")
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    try:
        torch.manual_seed(5)
        net = _Tail(ops)
        g = torch.Generator().manual_seed(11)
        data = torch.randn(2, 2, 8, 6, generator=g)     # [step][rank][batch][feat]
        tgt = torch.randn(2, 2, 8, 2, generator=g)
        params = list(net.parameters())
        results = []
        for early in (None, net.early_gradient_parameters()):
            red = dd.FlatGradReducer(params, gather=True, early_params=early)
            got = []
            for step, depth in enumerate([1, 2]):
                net.depth = depth
                red.prepare()
                for sub in range(2):                     # two accumulation sub-steps: only the last may start the exchange
                    if sub == 1:
                        red.arm()
                    nn.functional.mse_loss(net(data[step, rank]), tgt[step, rank]).backward()
                    if early is not None:
                        assert (red._early is not None) == (sub == 1), "early exchange started in the wrong pass"
                red.reduce()
                got.append([None if p.grad is None else p.grad.clone() for p in params])
            red.remove()
            results.append(got)
        for a_step, b_step in zip(*results):
            for a, b in zip(a_step, b_step):
                assert (a is None) == (b is None)
                if a is not None:
                    assert torch.equal(a, b), "two-bucket exchange != single bucket"
        assert results[0][0][2] is None and results[0][1][2] is not None     # body skipped at depth 1
        open(os.path.join(out_dir, "ok%d" % rank), "w").write("ok")
    finally:
        dist.destroy_process_group()


def test_two_bucket_overlapped_exchange_equals_single_bucket_gloo():
    """distributed.FlatGradReducer(early_params=...): the decoder tail's gradients are gathered and all-reduced from the
    backward milestone (ops.grad_milestone) while the rest of the backward pass runs; bit for bit the single bucket."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker_overlap, args=(2, os.path.join(d, "rendezvous"), d), nprocs=2, join=True)
        assert os.path.exists(os.path.join(d, "ok0")) and os.path.exists(os.path.join(d, "ok1"))
