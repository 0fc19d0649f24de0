"""state_dict manifest of the reference v3 generator and a deterministic weight generator.

The manifest reproduces, without importing the reference, the 958 (key, shape, dtype) entries of
`dec_ipt(opt).state_dict()` in registration order (reference
models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:104-388; LViT/GViT sub-keys v3:1106-1127, 1241-1262;
ActNorm buffer models/actnorm.py:16).  tests/test_manifest.py pins it against
tests/golden/state_manifest_*.txt, which tools/gen_golden.py dumped from the imported reference.

No real checkpoints exist offline (README.md:14 links a Baidu drive), so every parity test and the
benchmark use `generate_state_dict`: name + shape + seed -> tensor, identical on any machine with
the same torch build.  ActNorm parameters are generated explicitly with `initialized = 1`
(an uninitialised ActNorm holds garbage until its first forward, models/actnorm.py:12-13,25-37).
"""
import math
import zlib

import torch

from .config import NetConfig, BRANCHES


def _vit_entries(g, with_dead=True):
    D, H, S = g.dim, g.hidden, g.seq
    e = []
    if g.shrink > 1:       # v5 LViT: conv_shrink / conv_extend = Conv2d 1x1 + ActNorm2d (+ ReLU), registered first (v5:1101-1104)
        c, cm = g.channels, g.map_channels
        for nm, cout, cin in (("conv_shrink", c, cm), ("conv_extend", cm, c)):
            e += [(nm + ".0.weight", (cout, cin, 1, 1)), (nm + ".0.bias", (cout,)), (nm + ".1.weight", (cout,)), (nm + ".1.bias", (cout,)),
                  (nm + ".1.initialized", ())]
    e += [
        ("linear_encoding.weight", (D, D)), ("linear_encoding.bias", (D,)),
        ("mlp_head.0.weight", (H, D)), ("mlp_head.0.bias", (H,)),
        ("mlp_head.3.weight", (D, H)), ("mlp_head.3.bias", (D,)),
    ]
    if with_dead:
        e.append(("query_embed.weight", (1, D * S)))           # never used in forward (v3:1145,1158)
    e += [
        ("encoder.layers.0.self_attn.in_proj_weight", (3 * D, D)),
        ("encoder.layers.0.self_attn.out_proj.weight", (D, D)),
        ("encoder.layers.0.linear1.weight", (H, D)), ("encoder.layers.0.linear1.bias", (H,)),
        ("encoder.layers.0.linear2.weight", (D, H)), ("encoder.layers.0.linear2.bias", (D,)),
        ("encoder.layers.0.norm1.weight", (D,)), ("encoder.layers.0.norm1.bias", (D,)),
        ("encoder.layers.0.norm2.weight", (D,)), ("encoder.layers.0.norm2.bias", (D,)),
    ]
    if with_dead:                                              # TransformerDecoder: constructed, never called
        e += [
            ("decoder.layers.0.self_attn.in_proj_weight", (3 * D, D)),
            ("decoder.layers.0.self_attn.out_proj.weight", (D, D)),
            ("decoder.layers.0.multihead_attn.in_proj_weight", (3 * D, D)),
            ("decoder.layers.0.multihead_attn.out_proj.weight", (D, D)),
            ("decoder.layers.0.linear1.weight", (H, D)), ("decoder.layers.0.linear1.bias", (H,)),
            ("decoder.layers.0.linear2.weight", (D, H)), ("decoder.layers.0.linear2.bias", (D,)),
            ("decoder.layers.0.norm1.weight", (D,)), ("decoder.layers.0.norm1.bias", (D,)),
            ("decoder.layers.0.norm2.weight", (D,)), ("decoder.layers.0.norm2.bias", (D,)),
            ("decoder.layers.0.norm3.weight", (D,)), ("decoder.layers.0.norm3.bias", (D,)),
        ]
    e.append(("position_encoding.position_ids", (1, S)))
    e.append(("position_encoding.pe.weight", (S, D)))
    return e


def state_manifest(cfg: NetConfig, with_dead=True):
    """[(key, shape, torch.dtype)] in the reference's state_dict order."""
    nf = cfg.n_feats
    h = cfg.head_channels
    cfs = cfg.full_res              # networks_iid_hlgvit_crs_gd4_cfs.py / ..._crs_gd4.py: no ds_conv_e01 / us_conv_d01*, tail_color shared by R and D
    crs = cfg.variant == "crs"      # networks_iid_hlgvit_crs_gd4.py:327-330,364: D's skip fuse is a 1x1 conv over three maps, + a never-called SpatialPyramid
    out = []

    def add(k, shape, dtype=torch.float32):
        out.append((k, tuple(shape), dtype))

    if with_dead:
        add("sub_mean.weight", (3, 3, 1, 1)); add("sub_mean.bias", (3,))      # common.py:16-26, never called
        add("add_mean.weight", (3, 3, 1, 1)); add("add_mean.bias", (3,))
    add("head.0.0.weight", (h, cfg.n_colors, 5, 5)); add("head.0.0.bias", (h,))
    add("head.0.1.body.0.weight", (h, h, 3, 3)); add("head.0.1.body.0.bias", (h,))
    add("head.0.1.body.2.weight", (h, h, 3, 3)); add("head.0.1.body.2.bias", (h,))
    for g in cfg.vit_instances():
        for k, shape in _vit_entries(g, with_dead):
            add(g.name + "." + k, shape, torch.int64 if k.endswith(("position_ids", ".initialized")) else torch.float32)

    def conv_an(name, cout, cin, k=1):
        add(name + ".0.weight", (cout, cin, k, k)); add(name + ".0.bias", (cout,))
        add(name + ".1.weight", (cout,)); add(name + ".1.bias", (cout,))
        add(name + ".1.initialized", (), torch.int64)

    lg = {1: (nf, 2 * nf), 2: (2 * nf, 4 * nf), 3: (4 * nf, 8 * nf)}
    for l in (1, 2, 3):
        conv_an("lgcat_conv_e0%d" % l, *lg[l])
    for b in BRANCHES:
        for l in (3, 2, 1):
            conv_an("lgcat_conv_d0%d%s" % (l, b), *lg[l])
    add("ds_conv_e02.0.weight", (2 * nf, nf, 3, 3)); add("ds_conv_e02.0.bias", (2 * nf,))
    add("ds_conv_e03.0.weight", (4 * nf, 2 * nf, 3, 3)); add("ds_conv_e03.0.bias", (4 * nf,))
    if not cfs:
        add("ds_conv_e01.0.weight", (nf, h, 3, 3)); add("ds_conv_e01.0.bias", (nf,))
    for b in BRANCHES:
        add("us_conv_d03%s.0.weight" % b, (4 * nf, 2 * nf, 4, 4)); add("us_conv_d03%s.0.bias" % b, (2 * nf,))
        for nm, cin, cout in ((("us_conv_d02" + b, 2 * nf, nf),) if cfs else (("us_conv_d02" + b, 2 * nf, nf), ("us_conv_d01" + b, nf, h))):
            add(nm + ".0.weight", (cin, cout, 4, 4)); add(nm + ".0.bias", (cout,))
            add(nm + ".1.weight", (cout,)); add(nm + ".1.bias", (cout,))
            add(nm + ".1.initialized", (), torch.int64)
    for b in ("r", "s"):
        conv_an("sk_conv_d03" + b, 2 * nf, 4 * nf)
        conv_an("sk_conv_d02" + b, nf, 2 * nf)
    if crs:
        conv_an("sk_conv_d03d", 2 * nf, 6 * nf)
        conv_an("sk_conv_d02d", nf, 3 * nf)
    for nm, c in (() if crs else (("cfsm2g_d03d.0", 2 * nf), ("cfsm2g_d02d.0", nf))):
        bk = c // 4
        for fc in ("fc_avg_cf1", "fc_avg_cf2", "fc_max_cf1", "fc_max_cf2"):
            add("%s.%s.0.weight" % (nm, fc), (bk, c, 1, 1))
            add("%s.%s.2.weight" % (nm, fc), (c, bk, 1, 1))
    gray = "tail_gray" if cfs else "tail_S"
    for t, cout in ((("tail_color", cfg.n_colors),) if cfs else (("tail_R", cfg.n_colors), ("tail_D", cfg.n_colors))):
        add(t + ".0.1.weight", (h, h, 3, 3)); add(t + ".0.1.bias", (h,))
        add(t + ".0.2.weight", (h,)); add(t + ".0.2.bias", (h,)); add(t + ".0.2.initialized", (), torch.int64)
        add(t + ".0.5.weight", (cout, h, 7, 7)); add(t + ".0.5.bias", (cout,))
    add(gray + ".0.1.weight", (h, h, 3, 3)); add(gray + ".0.1.bias", (h,))
    add(gray + ".0.4.weight", (1, h, 7, 7)); add(gray + ".0.4.bias", (1,))
    if crs and with_dead:           # SpatialPyramid (crs:1441-1465), constructed at crs:364, its only call is commented out (crs:985)
        add("sp.refine1.weight", (32, 10, 3, 3)); add("sp.refine1.bias", (32,))
        add("sp.refine2.weight", (32, 32, 3, 3)); add("sp.refine2.bias", (32,))
        for k in ("1010", "1020", "1030", "1040", "1050"):
            add("sp.conv%s.weight" % k, (16, 32, 1, 1)); add("sp.conv%s.bias" % k, (16,))
        add("sp.refine3.0.weight", (3, 112, 3, 3)); add("sp.refine3.0.bias", (3,))
        for k, c in (("batch20", 20), ("batch1", 1)):
            add("sp.%s.weight" % k, (c,)); add("sp.%s.bias" % k, (c,)); add("sp.%s.initialized" % k, (), torch.int64)
    return out


def _is_dead(key):
    return (".decoder." in key or key.endswith("query_embed.weight")
            or key.startswith("sub_mean.") or key.startswith("add_mean.") or key.startswith("sp."))


def _gen_reference_init(key, shape, g):
    """What `define_G` leaves in a fresh reference module (init_weights v3:49-74 with init_type 'kaiming' + nn defaults):
    every Conv / ConvTranspose / Linear weight kaiming_normal_(a=0, mode='fan_in') with torch's fan_in = size(1) * k*k,
    their biases 0, LayerNorm 1 / 0, MHA in_proj kaiming_uniform_(a=sqrt(5)) = U(+-1/sqrt(fan_in)) (v3:1377), embeddings N(0,1)
    (v3:1330).  ActNorm weight / bias are uninitialised memory there (models/actnorm.py:12-13): zeros here, `initialized` = 0."""
    leaf = key.rsplit(".", 1)[-1]
    if len(shape) >= 2:
        if ".pe.weight" in key or key.endswith("query_embed.weight"):
            return torch.randn(shape, generator=g)
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "in_proj_weight" in key:
            return (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        return torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
    if ".norm" in key and leaf == "weight":
        return torch.ones(shape)
    return torch.zeros(shape)


def _gen(key, shape, dtype, seed, mode="trained"):
    g = torch.Generator()
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    if dtype == torch.int64:
        if key.endswith("position_ids"):
            return torch.arange(shape[1]).expand(1, -1).clone()
        return torch.tensor(0 if mode == "reference_init" else 1)   # ActNorm `initialized`
    leaf = key.rsplit(".", 1)[-1]
    if mode == "reference_init" and not (key.startswith("sub_mean") or key.startswith("add_mean")):
        return _gen_reference_init(key, shape, g)
    if key.startswith("sub_mean") or key.startswith("add_mean"):
        # common.py:16-26 MeanShift constants (dead in forward); keep the reference's values
        if leaf == "weight":
            return torch.eye(3).view(3, 3, 1, 1).clone()
        sign = -1.0 if key.startswith("sub_mean") else 1.0
        return sign * 255.0 * torch.tensor([0.4488, 0.4371, 0.4040])
    if len(shape) >= 2:
        if ".pe.weight" in key:
            # nn.Embedding default is N(0,1) (v3:1330); scaled so fp16 storage keeps ~3 digits after LN
            return torch.randn(shape, generator=g) * 0.5
        fan_in = 1
        for s in shape[1:]:
            fan_in *= s
        if "us_conv" in key:                                      # ConvTranspose2d weight is (Cin, Cout, k, k)
            fan_in = shape[0] * shape[2] * shape[3] // 4          # 2x2 taps hit each output pixel
        std = math.sqrt(2.0 / fan_in)
        if "in_proj_weight" in key:
            std = math.sqrt(1.0 / fan_in)
        if (".mlp_head.3." in key or ".linear2." in key or ".linear_encoding." in key or "out_proj" in key):
            std = 0.5 * math.sqrt(1.0 / fan_in)                    # residual branches: keep the stream O(1)
        if key.startswith("tail_") and shape[-1] == 7:
            std = 0.35 * math.sqrt(1.0 / fan_in)                   # keep tanh out of saturation
        return torch.randn(shape, generator=g) * std
    # 1-D parameters
    if ".norm" in key and leaf == "weight":
        return 1.0 + 0.1 * torch.randn(shape, generator=g)
    if ".norm" in key and leaf == "bias":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "weight":                                          # ActNorm log-scale (exp(w) ~ 0.6: a trained
        return -0.5 + 0.1 * torch.randn(shape, generator=g)       # ActNorm whitens its input, actnorm.py:25-37)
    return 0.05 * torch.randn(shape, generator=g)                 # biases (conv, linear, ActNorm)


def generate_state_dict(cfg: NetConfig, seed=0, with_dead=True, dtype=torch.float32, mode="trained"):
    """Deterministic stand-in for a checkpoint, keyed exactly like the reference's.
    mode "trained": a distribution shaped like a trained net (damped residual branches, ActNorm initialised);
    mode "reference_init": the distribution `define_G` itself produces (see _gen_reference_init), ActNorm uninitialised."""
    sd = {}
    for key, shape, dt in state_manifest(cfg, with_dead=with_dead):
        t = _gen(key, shape, dt, seed, mode)
        if dt != torch.int64:
            t = t.to(dtype)
        sd[key] = t
    return sd


def synthetic_input(batch, cfg: NetConfig, seed0=0, dtype=torch.float32):
    """image i = torch.rand(3,H,W, seed i)*2-1, the range of Normalize(0.5,0.5)."""
    n = cfg.image_size
    xs = []
    for i in range(batch):
        g = torch.Generator()
        g.manual_seed(seed0 + i)
        xs.append(torch.rand(1, cfg.n_colors, n, n, generator=g) * 2 - 1)
    return torch.cat(xs, 0).to(dtype)
