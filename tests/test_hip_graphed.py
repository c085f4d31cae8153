"""hipGraph replay of the eval-mode forward (graphed.GraphedEval, used by SRRunManager.validate_batched /
eval_ofa_net_sr.py / bench.py --config c5): the replayed output equals the eager forward bit for bit, one capture per
(input shape, network state), and every way the weights can change leads to a fresh capture.  `-m gpu`."""
import pytest
import torch

from conftest import amd

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _net():
    nets, dop = amd("elastic_nn.networks"), amd("elastic_nn.modules.dynamic_op")
    dop.DynamicSeparableConv2d.KERNEL_TRANSFORM_MODE = 1
    torch.manual_seed(5)
    net = nets.OFAMobileNetS4(ks_list=[3, 5, 7], expand_ratio_list=[3, 4, 6], depth_list=[2, 3, 4],
                              pixelshuffle_depth_list=[1, 2])
    net.init_model("he_fout")
    for m in net.modules():          # non-trivial running statistics
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.uniform_(-0.2, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    net.to(DEV).eval()
    net.set_active_subnet(ks=7, e=6, d=2, pixel_d=2)
    return net


def _eager(net, x):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return net(x)


def test_graph_replay_equals_eager_and_tracks_the_network_state():
    G, C, ops = amd("graphed"), amd("_C"), amd("ops")
    net = _net()
    g = G.GraphedEval(net, autocast_dtype=torch.bfloat16)
    xs = [torch.rand(2, 3, 31, 45, device=DEV), torch.rand(1, 3, 48, 64, device=DEV)]
    with torch.no_grad():
        for x in xs:
            y = g(x)
            assert torch.equal(y, _eager(net, x)) and y.shape == (x.shape[0], 3, 4 * x.shape[2], 4 * x.shape[3])
        assert g.captures == 2
        # replays: no new capture, fresh inputs give fresh outputs, the fused kernels are what the graph holds
        for x in (torch.rand_like(xs[0]), torch.rand_like(xs[1])):
            C.reset_launch_counts()
            y = g(x)
            assert C.launch_count("") == 0          # nothing was launched from the host: the graph replayed
            assert torch.equal(y, _eager(net, x))
        assert g.captures == 2 and g.replays == 4

        # another active sub-network -> another graph
        net.set_active_subnet(ks=3, e=4, d=3, pixel_d=2)
        assert torch.equal(g(xs[0]), _eager(net, xs[0])) and g.captures == 3
        # a tracked in-place weight update (what an optimizer step does)
        net.blocks[1].mobile_inverted_conv.point_linear.conv.conv.weight.mul_(1.5)
        y = g(xs[0])
        assert g.captures == 4 and torch.equal(y, _eager(net, xs[0]))
        # weights rewritten through .data (invisible to the version counters): the package's own rewriters bump the epoch
        before = g(xs[0]).clone()
        net.re_organize_middle_weights(expand_ratio_stage=0)
        y = g(xs[0])
        assert g.captures == 5 and torch.equal(y, _eager(net, xs[0]))
        # a state dict loaded
        sd = {k: v.clone() for k, v in net.state_dict().items()}
        k0 = next(k for k in sd if k.endswith("conv.weight"))
        sd[k0] = sd[k0] * 0.5
        net.load_state_dict(sd)
        y = g(xs[0])
        assert g.captures == 6 and torch.equal(y, _eager(net, xs[0])) and not torch.equal(y, before)


def test_graphed_eval_refuses_training_state():
    G = amd("graphed")
    net = _net()
    g = G.GraphedEval(net, autocast_dtype=torch.bfloat16)
    x = torch.rand(1, 3, 16, 16, device=DEV)
    with pytest.raises(RuntimeError):
        g(x)                                   # grad mode
    net.train()
    with torch.no_grad(), pytest.raises(RuntimeError):
        g(x)


def test_call_many_one_graph_for_all_size_buckets():
    """GraphedEval.call_many: the forwards of several differently sized inputs (one evaluation pass) captured as ONE graph and
    replayed with one host call; outputs equal the eager forwards bit for bit, a second pass replays, new data flows through."""
    G = amd("graphed")
    net = _net()
    g = G.GraphedEval(net, autocast_dtype=torch.bfloat16)
    gen = torch.Generator().manual_seed(3)
    sizes = [(2, 3, 24, 40), (1, 3, 31, 18), (3, 3, 16, 16)]
    with torch.no_grad():
        for rep in range(2):
            xs = [torch.rand(s, generator=gen).to(DEV) for s in sizes]
            ys = g.call_many(xs)
            assert g.captures == 1 and g.replays == rep + 1
            for x, y in zip(xs, ys):
                assert torch.equal(y, _eager(net, x))
