"""bench.py's training step (TrainWorkload: the objects the timed region runs) against the network-level CPU oracle
(oracle/s4_port.py) on the same weights, inputs, sub-network seeds and Adam groups: the loss trajectory bench.py reports
as `final_loss` is the reference algorithm's.  `-m gpu`.  Reference: progressive_shrinking.py:152-203 (hot loop),
sr_run_manager.py:115-133,180-191 (Adam, weight-decay groups)."""
import random

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype,tol", [("f32", 2e-4), ("bf16", 2e-2)])
def test_bench_train_step_losses_match_the_oracle_port(dtype, tol):
    import bench
    from oracle import s4_port
    dev = torch.device("cuda", 0)
    M = bench.mods()
    wl = bench.TrainWorkload(M, "c3", dev, batch=2, lr_size=16, dtype=dtype)
    sd = {k: v.detach().cpu().clone() for k, v in wl.net.state_dict().items()}
    decay, no_decay = [], []
    for k, v in sd.items():
        if s4_port.is_param(k):
            v.requires_grad_(True)
            (no_decay if ("bn" in k or "bias" in k) else decay).append(v)
    opt = torch.optim.Adam([{"params": decay, "weight_decay": 3e-5}, {"params": no_decay, "weight_decay": 0}], lr=1e-3)
    arch = s4_port.Arch(ks_list=(3, 5, 7), expand_list=(6,), depth_list=(4,), pd_list=(2,))
    hr, lr = wl.hr.cpu(), wl.lr.cpu()
    ref, got = [], []
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        random.seed(bench.subnet_seed(i))
        arch.sample_active_subnet()
        loss = F.mse_loss(s4_port.s4_forward(sd, lr, arch, training=True), hr)
        loss.backward()
        opt.step()
        ref.append(float(loss))
        got.append(float(wl.step(i)))
    # the first loss is a pure forward; every Adam step then moves each weight by ~lr whatever the gradient's size, so a
    # rounding-level gradient difference (fp32 summation order) is amplified step by step: 1e-7 / 5e-6 / 2.5e-4 measured
    for i, (a, b) in enumerate(zip(got, ref)):
        assert abs(a - b) <= tol * (1 + 4 * i) * abs(b), (dtype, got, ref)


def test_config2_whole_step_graph_equals_eager_steps():
    """bench.py --config c2 replays ONE captured hipGraph per training step (the sub-network is fixed: forward, loss,
    backward on both streams, the side-stream join and the fused Adam are the same launches on the same buffers every
    step).  Three eager warm-up steps + three replays must leave exactly the weights of six eager steps."""
    import bench
    dev = torch.device("cuda", 0)
    M = bench.mods()

    def run(capture):
        wl = bench.TrainWorkload(M, "c2", dev, batch=4, lr_size=16, dtype="bf16", graph_step=True)
        assert wl.want_graph
        if capture:
            assert wl.capture(), "the whole-step capture failed"
            for i in range(3):
                wl.step(3 + i)
        else:
            for i in range(6):
                wl.step(i)
        torch.cuda.synchronize()
        return {k: v.detach().clone() for k, v in wl.net.state_dict().items()}, float(wl.step(9))

    a, la = run(False)
    b, lb = run(True)
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert la == lb
