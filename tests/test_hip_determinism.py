"""The HIP hot path has no atomics and fixed summation orders: a training step is bit-reproducible, and the side-stream
schedule of the composite block backward (weight gradients beside the input-gradient chain) changes timing only.
Each configuration runs in its own process because the library reads its switches once."""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, importlib, random, sys
sys.path.insert(0, %r)
import torch
import torch.nn.functional as F
amd = lambda m: importlib.import_module("ofa-for-super-resolution_amd." + m)
dop = amd("elastic_nn.modules.dynamic_op"); nets = amd("elastic_nn.networks")
dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
torch.manual_seed(0)
net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                          pixelshuffle_depth_list=[1, 2]).to("cuda:0").train()
lr = torch.rand(4, 3, 32, 32, device="cuda:0"); hr = torch.rand(4, 3, 128, 128, device="cuda:0")
h = hashlib.sha256()
for step in range(3):
    random.seed(step); net.sample_active_subnet(); net.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = net(lr)
    loss = F.mse_loss(out.float(), hr); loss.backward()
    amd("ops").flush_deferred()   # deferred weight gradients -> .grad (ops.py)
    h.update(out.detach().float().cpu().numpy().tobytes())
    for n, p in net.named_parameters():
        if p.grad is not None:
            h.update(n.encode()); h.update(p.grad.detach().cpu().numpy().tobytes())
    for n, b in net.named_buffers():
        h.update(b.detach().cpu().numpy().tobytes())
print("HASH", h.hexdigest())
''' % ROOT


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    out = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("HASH ")]
    assert lines, out.stdout[-2000:]
    return lines[-1].split()[1]


def test_step_is_bit_reproducible_and_stream_schedule_invariant():
    a = _run({"OFASR_MBCONV_SIDE_STREAM": "1"})
    b = _run({"OFASR_MBCONV_SIDE_STREAM": "1"})
    c = _run({"OFASR_MBCONV_SIDE_STREAM": "0"})
    d = _run({"OFASR_MBCONV_DEFER_JOIN": "0"})
    e = _run({"OFASR_MBCONV_SHARED_TMP": "1"})
    assert a == b, "two identical runs differ: the path is not deterministic"
    assert a == c, "the side-stream schedule changed results"
    assert a == d, "deferring the weight-gradient join to the end of backward changed results"
    assert a == e, "one shared backward scratch buffer (ordered by the library's overlap waits) changed results"


def test_bench_contract_line():
    """bench.py prints ONE JSON line with the driver's contract keys plus `roofline` and `cpu_baseline` (a short run)."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--cpu-steps", "1", "--cpu-images", "1"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "sr_training_images_per_sec_4x_64to256" and j["unit"] == "images/s"
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["dtype"] == "bf16" and j["data"] == "synthetic" and j["vs_baseline"] is None and j["higher_is_better"] is True
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and j["value"] > 100 * c["value"]
