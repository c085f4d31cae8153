"""Network-level CPU oracle: a functional torch-CPU (fp32) restatement of OFAMobileNetS4.

TEST INFRASTRUCTURE ONLY (see oracle/ofasr_oracle.c header): used by tests/ as the checker for
whole-network forward/backward and by bench.py's `cpu_baseline` leg (kind "port").  It never
backs the product path.

It is deliberately NOT a module tree: it walks a flat state dict that uses the reference's key
names (e.g. `blocks.3.mobile_inverted_conv.depth_conv.conv.7to5_matrix`) and calls plain ATen
ops, so that a bug in the product's module plumbing cannot hide in a shared class.

Reference semantics restated (paths relative to /root/reference):
  forward / stage gating        ofa/elastic_nn/networks/ofa_mbs4.py:142-178   (incl. quirk Q1)
  set_active_subnet             ofa/elastic_nn/networks/ofa_mbs4.py:263-293   (incl. quirk Q2)
  sample_active_subnet          ofa/elastic_nn/networks/ofa_mbs4.py:316-370
  MB block                      ofa/elastic_nn/modules/dynamic_layers.py:70-84,
                                ofa/imagenet_codebase/networks/proxyless_nets.py:44-51
  elastic ops                   ofa/elastic_nn/modules/dynamic_op.py:46-84,104-112,148-167
  static conv layers            ofa/layers.py:94-98,131-151;  ofa/utils.py:259-260 (PixelShuffle)
Pinned by tests/golden/s4_net.npz + s4_meta.json (tests/test_oracle_golden_net.py).
"""
import random

import torch
import torch.nn.functional as F

N_MB = 16
GROUPS = [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 10, 11], [12, 13, 14, 15], [16, 17]]


def make_divisible(v, divisor, min_val=None):
    if min_val is None:
        min_val = divisor
    new_v = max(min_val, int(v + divisor / 2) // divisor * divisor)
    if new_v < 0.9 * v:
        new_v += divisor
    return new_v


def _as_list(v, n):
    # like the reference's int2list (imagenet_codebase/utils/__init__.py:97-103): a list is
    # returned AS IS (so set_active_subnet mutates the caller's depth list, quirk Q2)
    if isinstance(v, list):
        return v
    return list(v) if isinstance(v, tuple) else [v] * n


class Arch(object):
    """the mutable 'active sub-network' state of the supernet"""

    def __init__(self, ks_list=(3, 5, 7), expand_list=(3, 4, 6), depth_list=(2, 3, 4), pd_list=(1, 2)):
        self.ks_list, self.expand_list = sorted(ks_list), sorted(expand_list)
        self.depth_list, self.pd_list = sorted(depth_list), sorted(pd_list)
        self.ks = [max(self.ks_list)] * N_MB
        self.e = [max(self.expand_list)] * N_MB
        self.runtime_depth = [len(g) for g in GROUPS]

    def set_active_subnet(self, ks=None, e=None, d=None, pixel_d=None):
        ks = _as_list(ks, N_MB)
        e = _as_list(e, N_MB)
        depth = _as_list(d, len(GROUPS) - 1)     # a caller-owned list is mutated, like the reference
        pd = _as_list(pixel_d, 1)
        depth.insert(-1, pd[0])                  # Q2: lands BEFORE the last element
        for i in range(N_MB):
            if ks[i] is not None:
                self.ks[i] = ks[i]
            if e[i] is not None:
                self.e[i] = e[i]
        for i, dd in enumerate(depth):
            if dd is not None:
                self.runtime_depth[i] = min(len(GROUPS[i]), dd)

    def sample_active_subnet(self):
        ks = [random.choice(self.ks_list) for _ in range(N_MB)]
        e = [random.choice(self.expand_list) for _ in range(N_MB)]
        d = [random.choice(self.depth_list) for _ in range(len(GROUPS) - 1)]
        pd = [random.choice(self.pd_list)]
        self.set_active_subnet(ks, e, d, pd)
        return {"wid": None, "ks": ks, "e": e, "d": d, "pixel_d": pd}


def _bn(sd, prefix, x, training, momentum, eps):
    C = x.shape[1]
    w, b = sd[prefix + "weight"], sd[prefix + "bias"]
    rm, rv = sd[prefix + "running_mean"], sd[prefix + "running_var"]
    if training:
        sd[prefix + "num_batches_tracked"] += 1
    # F.batch_norm updates the running-stat views in place (slices of the full buffers)
    return F.batch_norm(x, rm[:C], rv[:C], w[:C], b[:C], training, momentum if training else 0.0, eps)


def active_filter(sd, prefix, C, k, ks_list, transform):
    w = sd[prefix + "conv.weight"]                       # [Cmax,1,kmax,kmax]
    kmax = w.shape[-1]
    s = kmax // 2 - k // 2
    if (not transform) or k == kmax:
        return w[:C, :, s:s + k, s:s + k]
    cur, kc = w[:C], kmax
    for kt in sorted(ks_list, reverse=True)[1:]:
        if kt < k:
            break
        s = kc // 2 - kt // 2
        crop = cur[:, :, s:s + kt, s:s + kt].reshape(C, kt * kt)
        cur = F.linear(crop, sd[prefix + "%dto%d_matrix" % (kc, kt)]).view(C, 1, kt, kt)
        kc = kt
    return cur


def _mb_block(sd, p, x, k, e, ks_list, transform, training, momentum, eps):
    cin = x.shape[1]
    mid = make_divisible(round(cin * e), 8)
    w1 = sd[p + "inverted_bottleneck.conv.conv.weight"]
    h = F.conv2d(x, w1[:mid, :cin])
    h = F.relu6(_bn(sd, p + "inverted_bottleneck.bn.bn.", h, training, momentum, eps))
    f = active_filter(sd, p + "depth_conv.conv.", mid, k, ks_list, transform)
    h = F.conv2d(h, f, None, 1, k // 2, 1, mid)
    h = F.relu6(_bn(sd, p + "depth_conv.bn.bn.", h, training, momentum, eps))
    w2 = sd[p + "point_linear.conv.conv.weight"]
    cout = w2.shape[0]
    h = F.conv2d(h, w2[:cout, :mid])
    h = _bn(sd, p + "point_linear.bn.bn.", h, training, momentum, eps)
    return h + x


def _conv_layer(sd, p, x, training, momentum, eps, shuffle=False):
    w = sd[p + "conv.weight"]
    h = F.conv2d(x, w, None, 1, w.shape[-1] // 2)
    h = _bn(sd, p + "bn.", h, training, momentum, eps)
    return F.pixel_shuffle(h, 2) if shuffle else h


def s4_forward(sd, x, arch, training, transform=True, momentum=0.1, eps=1e-5, compat=True):
    """sd: {reference key: tensor}; BN buffers are updated in place when training.
    compat=True reproduces the as-committed stage indexing; compat=False uses
    runtime_depth[4] for the shuffle stage."""
    h = _conv_layer(sd, "dec_first_conv_block.", x, training, momentum, eps)
    skip = h
    for stage in range(4):
        for idx in GROUPS[stage][:arch.runtime_depth[stage]]:
            h = _mb_block(sd, "blocks.%d.mobile_inverted_conv." % idx, h, arch.ks[idx], arch.e[idx],
                          arch.ks_list, transform, training, momentum, eps)
    h = _conv_layer(sd, "dec_final_conv_blocks.0.", h, training, momentum, eps) + skip
    h = _conv_layer(sd, "dec_final_conv_blocks.1.", h, training, momentum, eps)
    d_shuffle = arch.runtime_depth[0] if compat else arch.runtime_depth[4]
    # the net holds max(pixelshuffle_depth_list) conv+PixelShuffle blocks (ofa_mbs4.py:111-120): a 2x-only supernet
    # (pixelshuffle_depth_list=[1], BASELINE configs 1 and 2) has no blocks.17
    shuffle_group = [i for i in GROUPS[4] if ("blocks.%d.conv.weight" % i) in sd]
    for idx in shuffle_group[:d_shuffle]:
        h = _conv_layer(sd, "blocks.%d." % idx, h, training, momentum, eps, shuffle=True)
    return _conv_layer(sd, "dec_final_output_conv_block.", h, training, momentum, eps)


def state_dict_shapes(ks_max=7, e_max=6, transform=True, n_shuffle=2):
    """the reference's 356-entry state dict layout for the full S4 supernet."""
    shapes = {}

    def bn(p, c):
        shapes[p + "weight"] = (c,)
        shapes[p + "bias"] = (c,)
        shapes[p + "running_mean"] = (c,)
        shapes[p + "running_var"] = (c,)
        shapes[p + "num_batches_tracked"] = ()

    mid = round(64 * e_max)
    for i in range(N_MB):
        p = "blocks.%d.mobile_inverted_conv." % i
        shapes[p + "inverted_bottleneck.conv.conv.weight"] = (mid, 64, 1, 1)
        bn(p + "inverted_bottleneck.bn.bn.", mid)
        if transform:
            shapes[p + "depth_conv.conv.5to3_matrix"] = (9, 9)
            shapes[p + "depth_conv.conv.7to5_matrix"] = (25, 25)
        shapes[p + "depth_conv.conv.conv.weight"] = (mid, 1, ks_max, ks_max)
        bn(p + "depth_conv.bn.bn.", mid)
        shapes[p + "point_linear.conv.conv.weight"] = (64, mid, 1, 1)
        bn(p + "point_linear.bn.bn.", 64)
    for i in (16, 17)[:n_shuffle]:
        shapes["blocks.%d.conv.weight" % i] = (256, 64, 5, 5)
        bn("blocks.%d.bn." % i, 256)
    shapes["dec_first_conv_block.conv.weight"] = (64, 3, 5, 5)
    bn("dec_first_conv_block.bn.", 64)
    for i in (0, 1):
        shapes["dec_final_conv_blocks.%d.conv.weight" % i] = (64, 64, 5, 5)
        bn("dec_final_conv_blocks.%d.bn." % i, 64)
    shapes["dec_final_output_conv_block.conv.weight"] = (3, 64, 5, 5)
    bn("dec_final_output_conv_block.bn.", 3)
    return shapes


def he_fout_state_dict(seed=0, **kw):
    """init_model('he_fout') (ofa/utils.py:134-155): conv ~ N(0, sqrt(2/(k*k*Cout))), BN 1/0;
    transform matrices identity (dynamic_op.py:40).  Returns float32 torch tensors."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in state_dict_shapes(**kw).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_var") or (name.endswith("weight") and len(shape) == 1):
            sd[name] = torch.ones(shape)
        elif name.endswith("running_mean") or name.endswith("bias"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("_matrix"):
            sd[name] = torch.eye(shape[0])
        else:
            n = shape[2] * shape[3] * shape[0]
            sd[name] = torch.randn(shape, generator=g) * (2.0 / n) ** 0.5
    return sd


def is_param(name):
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked"))
