"""Eval-mode forward of an SR network replayed from hipGraphs.

The inference path of a sampled sub-network on a Set14-sized image is ~20 short kernels per forward call (one per MB
block, one per ConvLayer); issued one by one from Python the host needs ~30 us per launch and the GPU waits (measured
on BASELINE config 5: 6.9 ms per pass of host enqueue time against ~3.5 ms of kernels, tools/host_profile_eval.py).
`GraphedEval(net)` captures the whole forward once per input shape (torch.cuda.CUDAGraph = hipGraph on ROCm; every
kernel of csrc/ is launched on the capturing stream with no synchronisation, allocation or event inside the library)
and replays it with one host call.

A captured graph holds raw device pointers -- the weights, the prepared inference operands (ops.py operand cache) and
its private activation pool -- so a graph is only replayed while the network is in the state it was captured in:
the key is (input shape / dtype, autocast dtype, the active sub-network's description, and (address, version) of
every parameter and buffer).  Any optimizer step, load_state_dict, re-organisation or set_active_subnet therefore
leads to a fresh capture; graphs of an earlier operand-cache epoch (ops.clear_infer_cache) are dropped at once,
the rest least recently used first.

Mirror of nothing in the reference (its eval loop, eval_ofa_net_sr.py:187-220, calls the module eagerly); used by
SRRunManager.validate_batched, eval_ofa_net_sr.py and bench.py --config c5.
"""
import collections

import torch

from . import ops


class GraphedEval(object):
    def __init__(self, net, autocast_dtype=None, max_graphs=32, copy_output=True):
        self.net = net
        self.autocast_dtype = autocast_dtype
        self.max_graphs = max_graphs
        self.copy_output = copy_output
        self._graphs = collections.OrderedDict()
        self._epoch = ops.infer_epoch()
        self._tensors = self._bns = None
        self.captures = 0
        self.replays = 0

    # ------------------------------------------------------------------ state the captured pointers depend on
    def _state(self):
        net = self.net
        if self._tensors is None:     # the module tree is fixed: walk it once (this runs on every call)
            self._tensors = list(net.parameters()) + list(net.buffers())
            self._bns = [m for m in net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)]
        tensors = tuple((t.data_ptr(), t._version) for t in self._tensors)
        try:
            arch = net.module_str
        except (AttributeError, NotImplementedError):
            arch = None
        # BN eps is folded into the prepared operands but is a plain float (set_bn_param rewrites it without touching
        # any tensor); the two switches select which kernels a forward launches
        eps = tuple(m.eps for m in self._bns)
        # the operand cache's epoch: bumped by everything that rewrites weights behind the version counters
        return arch, hash(tensors), hash(eps), ops.FUSED_INFER, ops.INFER_CACHE, ops.infer_epoch()

    def _forward(self, x):
        if self.autocast_dtype is None:
            return self.net(x)
        with torch.autocast("cuda", dtype=self.autocast_dtype):
            return self.net(x)

    def _capture(self, x):
        static_in = x.clone()
        side = torch.cuda.Stream(device=x.device)
        side.wait_stream(torch.cuda.current_stream(x.device))
        with torch.cuda.stream(side):       # warm-up off the capturing stream: lazy initialisation, operand preparation,
            for _ in range(2):              # allocator growth and the library's launch-site registration happen here
                self._forward(static_in)
        torch.cuda.current_stream(x.device).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            static_out = self._forward(static_in)
        self.captures += 1
        # the prepared inference operands the graph points at stay alive as long as the graph does, whatever the cache does
        held = ops.infer_operand_buffers()
        return graph, static_in, static_out, held

    def __call__(self, x):
        if torch.is_grad_enabled():
            raise RuntimeError("GraphedEval replays an inference graph: call it under torch.no_grad()")
        if self.net.training or any(m.training for m in self.net.modules() if isinstance(m, torch.nn.modules.batchnorm._BatchNorm)):
            raise RuntimeError("GraphedEval needs the network (and its BatchNorm layers) in eval mode")
        if not x.is_cuda:
            raise RuntimeError("GraphedEval needs a GPU tensor")
        if self._epoch != ops.infer_epoch():
            # weights were rewritten behind the version counters (or the net went through .train()): no graph captured
            # before can match again -- free their activation pools and operand buffers now rather than by LRU
            self._graphs.clear()
            self._epoch = ops.infer_epoch()
        key = (tuple(x.shape), x.dtype, str(x.device), self.autocast_dtype) + self._state()
        entry = self._graphs.get(key)
        if entry is None:
            entry = self._graphs[key] = self._capture(x)
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        graph, static_in, static_out = entry[:3]
        static_in.copy_(x)
        graph.replay()
        self.replays += 1
        return static_out.clone() if self.copy_output else static_out

    def call_many(self, xs):
        """the forward of SEVERAL inputs (the size buckets of one evaluation pass) as ONE captured graph and one replay:
        a replay costs the host ~0.1 ms whatever is in the graph, so a Set14 pass of 10 buckets is replay-bound when each
        bucket has a graph of its own.  Same keying as __call__ (all shapes + the network state)."""
        if torch.is_grad_enabled():
            raise RuntimeError("GraphedEval replays an inference graph: call it under torch.no_grad()")
        if self.net.training:
            raise RuntimeError("GraphedEval needs the network (and its BatchNorm layers) in eval mode")
        if self._epoch != ops.infer_epoch():
            self._graphs.clear()
            self._epoch = ops.infer_epoch()
        key = ("many", tuple((tuple(x.shape), x.dtype, str(x.device)) for x in xs), self.autocast_dtype) + self._state()
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [x.clone() for x in xs]
            dev = xs[0].device
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                for _ in range(2):
                    for si in static_in:
                        self._forward(si)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = [self._forward(si) for si in static_in]
            self.captures += 1
            entry = self._graphs[key] = (graph, static_in, static_out, ops.infer_operand_buffers())
            while len(self._graphs) > self.max_graphs:
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        graph, static_in, static_out = entry[:3]
        torch._foreach_copy_(static_in, list(xs))
        graph.replay()
        self.replays += 1
        return [o.clone() for o in static_out] if self.copy_output else list(static_out)

    def clear(self):
        self._graphs.clear()
