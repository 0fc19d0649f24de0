"""CPU oracle for the CFEN-ViT v3 generator forward.   *** TEST INFRASTRUCTURE ONLY ***

A from-scratch functional restatement (plain tensor arithmetic on CPU, fp32 or fp64) of
`dec_ipt.forward` in the reference, models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:392-1020, taking
the reference's own state_dict keys.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this file; the product path (cfen_vit_dehazing_amd/) never does.

PINNING: the reference has no tests or golden vectors for this path, so the oracle is
pinned against outputs of the reference itself: tools/gen_golden.py imports
/root/reference/models/networks_iid_hlgvit_crs_gd4_cfs_v3.py in the build container, runs it on
seeded weights/inputs and commits the vectors under tests/golden/; tests/test_oracle_golden.py checks
this file against them (<= 2e-5 max-abs in fp32, full net and per block).

Differences of *form* (not of arithmetic) from the reference:
  * the recursive Crop2x2/Join2x2 quadrant split (v3:1025-1056, 403-486) is one reshape into
    non-overlapping windows, and all windows go through the block as one batch (the reference's
    per-window Python loop is batch-invariant: SURVEY Appendix A);
  * F.unfold/F.fold with kernel==stride (v3:1140,1186) are index permutations;
  * nn.MultiheadAttention(bias=False) (v3:1364) is written out as QK^T/softmax/PV;
  * bilinear x2 (align_corners=False) and 2x2 average pooling are written out explicitly.
"""
import math

import torch
import torch.nn.functional as F

BRANCHES = ("r", "s", "d")


# --------------------------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------------------------
def window_partition(x, ws):
    """(B,C,H,W) -> (B*nwy*nwx, C, ws, ws); the leaf order of nested Crop2x2 is irrelevant because
    Join2x2 writes every window back where it came from (v3:1046-1056)."""
    B, C, H, W = x.shape
    x = x.reshape(B, C, H // ws, ws, W // ws, ws).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(B * (H // ws) * (W // ws), C, ws, ws)


def window_merge(xw, B, H, W):
    n, C, ws, _ = xw.shape
    x = xw.reshape(B, H // ws, W // ws, C, ws, ws).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(B, C, H, W)


def unfold_tokens(x, p):
    """F.unfold(x, p, stride=p).transpose(1,2): (N,C,H,W) -> (N, S, C*p*p); feature = c*p*p + i*p + j,
    token = row*(W/p) + col (v3:1140)."""
    N, C, H, W = x.shape
    x = x.reshape(N, C, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5)
    return x.reshape(N, (H // p) * (W // p), C * p * p)


def fold_tokens(t, C, H, W, p):
    """Inverse of unfold_tokens == F.fold(..., kernel=stride=p) (v3:1186)."""
    N = t.shape[0]
    x = t.reshape(N, H // p, W // p, C, p, p).permute(0, 3, 1, 4, 2, 5)
    return x.reshape(N, C, H, W)


def layer_norm(x, w, b, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d(affine=False, track_running_stats=False): biased variance (v3:292-302)."""
    mu = x.mean((2, 3), keepdim=True)
    var = ((x - mu) ** 2).mean((2, 3), keepdim=True)
    return (x - mu) / torch.sqrt(var + eps)


def actnorm(sd, prefix, x):
    """models/actnorm.py:22-42: y = (x + bias) * exp(weight); when `initialized` is 0 the parameters are first filled from this
    batch (actnorm.py:25-37) and written back into `sd`, exactly as the reference module mutates itself on its first forward."""
    if int(sd[prefix + ".initialized"]) != 1:
        w, b = actnorm_init_params(x)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = w, b
        sd[prefix + ".initialized"] = torch.tensor(1)
    return (x + sd[prefix + ".bias"].view(1, -1, 1, 1)) * torch.exp(sd[prefix + ".weight"]).view(1, -1, 1, 1)


def actnorm_init_params(x):
    """models/actnorm.py:25-37 data-dependent first-call init: returns (weight, bias)."""
    c = x.shape[1]
    xt = x.transpose(0, 1).contiguous().view(c, -1)
    mean = xt.mean(1)
    var = xt.var(1)                                       # unbiased
    var = torch.clamp(var, min=0.2)
    return -0.5 * torch.log(var), -mean


def avgpool2(x):
    N, C, H, W = x.shape
    return x.reshape(N, C, H // 2, 2, W // 2, 2).mean((3, 5))


def upsample2_bilinear(x):
    """nn.Upsample(scale_factor=2, mode='bilinear') => align_corners=False (v3:117,1238).
    src = (dst + 0.5)/2 - 0.5 clamped at 0; i1 = min(i0+1, n-1)."""
    def axis(n):
        d = torch.arange(2 * n, dtype=x.dtype)
        s = torch.clamp((d + 0.5) * 0.5 - 0.5, min=0)
        i0 = s.floor().long()
        i1 = torch.clamp(i0 + 1, max=n - 1)
        w1 = s - i0.to(x.dtype)
        return i0, i1, w1
    N, C, H, W = x.shape
    y0, y1, wy = axis(H)
    x0, x1, wx = axis(W)
    rows = x[:, :, y0, :] * (1 - wy).view(1, 1, -1, 1) + x[:, :, y1, :] * wy.view(1, 1, -1, 1)
    return rows[:, :, :, x0] * (1 - wx).view(1, 1, 1, -1) + rows[:, :, :, x1] * wx.view(1, 1, 1, -1)


# --------------------------------------------------------------------------------------------
# transformer block shared by LViT and GViT  (v3:1136-1189, 1272-1325, 1382-1390)
# --------------------------------------------------------------------------------------------
def vit_tokens(sd, prefix, tok, heads, taps=None):
    """tok: (N, S, D) unfolded tokens -> (N, S, D) after embed, +pos, encoder layer, mlp_head."""
    N, S, D = tok.shape
    dh = D // heads
    x = tok @ sd[prefix + ".linear_encoding.weight"].t() + sd[prefix + ".linear_encoding.bias"] + tok   # v3:1143
    x = x + sd[prefix + ".position_encoding.pe.weight"][:S].unsqueeze(0)                                   # v3:1152,1166
    if taps is not None:
        taps["embed"] = x
    e = prefix + ".encoder.layers.0"
    y = layer_norm(x, sd[e + ".norm1.weight"], sd[e + ".norm1.bias"])                                    # v3:1383
    qkv = y @ sd[e + ".self_attn.in_proj_weight"].t()                                                    # no bias (v3:1364)
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(N, S, heads, dh).transpose(1, 2)
    k = k.reshape(N, S, heads, dh).transpose(1, 2)
    v = v.reshape(N, S, heads, dh).transpose(1, 2)
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1) @ v
    a = a.transpose(1, 2).reshape(N, S, D) @ sd[e + ".self_attn.out_proj.weight"].t()
    x = x + a                                                                                            # v3:1386
    if taps is not None:
        taps["attn"] = x
    y = layer_norm(x, sd[e + ".norm2.weight"], sd[e + ".norm2.bias"])                                    # v3:1387
    y = torch.relu(y @ sd[e + ".linear1.weight"].t() + sd[e + ".linear1.bias"])
    x = x + y @ sd[e + ".linear2.weight"].t() + sd[e + ".linear2.bias"]                                  # v3:1388-1389
    if taps is not None:
        taps["ffn"] = x
    y = torch.relu(x @ sd[prefix + ".mlp_head.0.weight"].t() + sd[prefix + ".mlp_head.0.bias"])
    x = x + y @ sd[prefix + ".mlp_head.3.weight"].t() + sd[prefix + ".mlp_head.3.bias"]                  # v3:1173
    return x


def lvit(sd, prefix, x, heads, ws, p=2, shrink=False):
    """All windows of one LViT instance at once. x: (B,C,H,W).
    shrink (networks_iid_hlgvit_crs_gd4_cfs_v5.py:1139,1190): Conv2d 1x1 C -> C/4 + ActNorm2d + ReLU in front of the tokens and the
    mirror-image conv_extend behind the fold.  Both are per-pixel once the ActNorm is initialised, so they commute with the window split;
    an UNINITIALISED ActNorm there takes its statistics from the module's first call = the first window of the reference's crop
    recursion (the top-left one, v5:403-410), all B images."""
    B, C, H, W = x.shape
    if shrink:
        y = F.conv2d(x, sd[prefix + ".conv_shrink.0.weight"], sd[prefix + ".conv_shrink.0.bias"])
        _actnorm_first_window(sd, prefix + ".conv_shrink.1", y, ws)
        x = torch.relu(actnorm(sd, prefix + ".conv_shrink.1", y))
        C = x.shape[1]
    xw = window_partition(x, ws)
    t = vit_tokens(sd, prefix, unfold_tokens(xw, p), heads)
    out = window_merge(fold_tokens(t, C, ws, ws, p), B, H, W)
    if shrink:
        y = F.conv2d(out, sd[prefix + ".conv_extend.0.weight"], sd[prefix + ".conv_extend.0.bias"])
        _actnorm_first_window(sd, prefix + ".conv_extend.1", y, ws)
        out = torch.relu(actnorm(sd, prefix + ".conv_extend.1", y))
    return out


def _actnorm_first_window(sd, prefix, y, ws):
    if int(sd[prefix + ".initialized"]) != 1:
        w, b = actnorm_init_params(y[:, :, :ws, :ws])
        sd[prefix + ".weight"], sd[prefix + ".bias"] = w, b
        sd[prefix + ".initialized"] = torch.tensor(1)


def gvit(sd, prefix, x, heads, p=4):
    """v3:1272-1325: pool/4, ViT with 4x4 patches over the whole pooled map, fold, bilinear x2 twice."""
    B, C, H, W = x.shape
    xp = avgpool2(avgpool2(x))
    t = vit_tokens(sd, prefix, unfold_tokens(xp, p), heads)
    y = fold_tokens(t, C, H // 4, W // 4, p)
    return upsample2_bilinear(upsample2_bilinear(y))


def cfsm2g(sd, prefix, x0, x1, x2):
    """v3:1481-1517."""
    comb = x0 + x1 + x2
    avg = comb.mean((2, 3), keepdim=True)
    mx = comb.amax((2, 3), keepdim=True)

    def fc(name, z):
        z = torch.relu(F.conv2d(z, sd["%s.%s.0.weight" % (prefix, name)]))
        return F.conv2d(z, sd["%s.%s.2.weight" % (prefix, name)])
    g1 = torch.sigmoid(fc("fc_avg_cf1", avg) + fc("fc_max_cf1", mx))
    g2 = torch.sigmoid(fc("fc_avg_cf2", avg) + fc("fc_max_cf2", mx))
    return x0 + x1 * g1 + x2 * g2


def conv_actnorm_relu(sd, prefix, x):
    """Sequential(Conv2d 1x1, ActNorm2d, ReLU) (v3:255-284, 329-338)."""
    return torch.relu(actnorm(sd, prefix + ".1", F.conv2d(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])))


def ds_conv(sd, prefix, x):
    """Conv2d 3x3 s2 p1 -> InstanceNorm -> ReLU (v3:292-298)."""
    return torch.relu(instance_norm(F.conv2d(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"], stride=2, padding=1)))


def us_conv(sd, prefix, x, norm):
    """ConvTranspose2d 4x4 s2 p1 -> InstanceNorm | ActNorm -> ReLU (v3:301-322)."""
    y = F.conv_transpose2d(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"], stride=2, padding=1)
    y = instance_norm(y) if norm == "in" else actnorm(sd, prefix + ".1", y)
    return torch.relu(y)


def head(sd, x):
    """conv5x5 + ResBlock (v3:123-127; common.py:11-14, 41-62)."""
    x = F.conv2d(x, sd["head.0.0.weight"], sd["head.0.0.bias"], padding=2)
    r = torch.relu(F.conv2d(x, sd["head.0.1.body.0.weight"], sd["head.0.1.body.0.bias"], padding=1))
    r = F.conv2d(r, sd["head.0.1.body.2.weight"], sd["head.0.1.body.2.bias"], padding=1)
    return r + x


def tail(sd, name, x):
    """v3:348-383 / cfs:334-358 (Upsampler is empty: common.py:64-81 with log2(1) = 0 stages)."""
    p = name + ".0"
    x = F.conv2d(x, sd[p + ".1.weight"], sd[p + ".1.bias"], padding=1)
    if name not in ("tail_S", "tail_gray"):
        x = actnorm(sd, p + ".2", x)
        last = p + ".5"
    else:
        last = p + ".4"
    x = torch.relu(x)
    x = F.pad(x, (3, 3, 3, 3), mode="reflect")
    return torch.tanh(F.conv2d(x, sd[last + ".weight"], sd[last + ".bias"]))


# --------------------------------------------------------------------------------------------
# whole network
# --------------------------------------------------------------------------------------------
def forward(sd, x, num_heads=4, patch_size=32, stages=None, variant="v3"):
    """dec_ipt.forward (v3:392-1020).  x: (B,3,H,W) in [-1,1] -> [xr (B,3,H,W), xs (B,1,H,W), xd (B,3,H,W)].
    If `stages` is a dict it receives the 58 top-level stage outputs named as in SURVEY Appendix D.
    variant "cfs" = models/networks_iid_hlgvit_crs_gd4_cfs.py:362-980: the same three levels run on the head's own (full-resolution)
    output -- no ds_conv_e01 / us_conv_d01* -- and the tails (tail_color shared by R and D, tail_gray for S) read `d_01 + xf` directly.
    variant "crs" = models/networks_iid_hlgvit_crs_gd4.py:366-987: as "cfs" with sk_conv_d03d / sk_conv_d02d (1x1 conv over the three
    decoders' upsampled maps) where cfs has CFSM2G.  variant "v5" = models/networks_iid_hlgvit_crs_gd4_cfs_v5.py: v3 whose LViT blocks
    sit between a 1x1 conv_shrink (C -> C/4) and conv_extend (C/4 -> C), v5:1139,1190."""
    ws = patch_size
    cfs = variant in ("cfs", "crs")
    crs = variant == "crs"          # networks_iid_hlgvit_crs_gd4.py:854,889: D's skip fuse is sk_conv_d0Xd over cat(D, R, S) instead of CFSM2G
    v5 = variant == "v5"            # networks_iid_hlgvit_crs_gd4_cfs_v5.py: v3 with the LViT blocks on a quarter of the channels

    def rec(name, t):
        if stages is not None:
            stages[name] = t
        return t

    def level(tag, l, xin):
        """LViT || GViT -> lgcat 1x1 -> + residual (v3:403-488 etc.)."""
        heads = num_heads << (l - 1)
        if tag == "e":
            ln, gn, cn = "localvit_encoder_0%d" % l, "globalvit_encoder_0%d" % l, "lgcat_conv_e0%d" % l
        else:
            ln, gn, cn = ("localvit_decoder_0%d%s" % (l, tag), "globalvit_decoder_0%d%s" % (l, tag),
                          "lgcat_conv_d0%d%s" % (l, tag))
        lo = rec(ln, lvit(sd, ln, xin, heads, ws, shrink=v5))
        gl = rec(gn, gvit(sd, gn, xin, heads))
        return rec(cn, conv_actnorm_relu(sd, cn, torch.cat((lo, gl), 1)) + xin)

    xh = rec("head", head(sd, x))
    xf = xh if cfs else rec("ds_conv_e01", ds_conv(sd, "ds_conv_e01", xh))
    x_e_01 = level("e", 1, xf)
    x_e_01_ds = rec("ds_conv_e02", ds_conv(sd, "ds_conv_e02", x_e_01))
    x_e_02 = level("e", 2, x_e_01_ds)
    x_e_02_ds = rec("ds_conv_e03", ds_conv(sd, "ds_conv_e03", x_e_02))
    x_e_03 = level("e", 3, x_e_02_ds)

    ups = {}
    outs = {}
    for t in BRANCHES:                                        # R, S, then D (D reads R's and S's upsampled maps)
        d3 = level(t, 3, x_e_03)
        u3 = rec("us_conv_d03" + t, us_conv(sd, "us_conv_d03" + t, d3, "in"))
        ups[(t, 3)] = u3
        if t == "d" and crs:
            in2 = rec("sk_conv_d03d", conv_actnorm_relu(sd, "sk_conv_d03d", torch.cat((u3, ups[("r", 3)], ups[("s", 3)]), 1)))   # crs:854
        elif t == "d":
            in2 = rec("cfsm2g_d03d", cfsm2g(sd, "cfsm2g_d03d.0", u3, ups[("r", 3)], ups[("s", 3)]))   # v3:885
        else:
            in2 = rec("sk_conv_d03" + t, conv_actnorm_relu(sd, "sk_conv_d03" + t, torch.cat((u3, x_e_02), 1)))
        d2 = level(t, 2, in2)
        u2 = rec("us_conv_d02" + t, us_conv(sd, "us_conv_d02" + t, d2, "an"))
        ups[(t, 2)] = u2
        if t == "d" and crs:
            in1 = rec("sk_conv_d02d", conv_actnorm_relu(sd, "sk_conv_d02d", torch.cat((u2, ups[("r", 2)], ups[("s", 2)]), 1)))   # crs:889
        elif t == "d":
            in1 = rec("cfsm2g_d02d", cfsm2g(sd, "cfsm2g_d02d.0", u2, ups[("r", 2)], ups[("s", 2)]))   # v3:920
        else:
            in1 = rec("sk_conv_d02" + t, conv_actnorm_relu(sd, "sk_conv_d02" + t, torch.cat((u2, x_e_01), 1)))
        d1 = level(t, 1, in1)
        tn = "tail_" + t.upper()
        if cfs:
            outs[t] = rec(tn, tail(sd, "tail_gray" if t == "s" else "tail_color", d1 + xf))              # cfs:669,823,977
        else:
            u1 = rec("us_conv_d01" + t, us_conv(sd, "us_conv_d01" + t, d1 + xf, "an"))                  # v3:696,852,1008
            outs[t] = rec(tn, tail(sd, tn, u1))
    return [outs["r"], outs["s"], outs["d"]]


# --------------------------------------------------------------------------------------------
# metrics used by the parity reports (the reference has no PSNR code; SSIM follows
# pytorch_msssim/__init__.py:19-70: 11x11 gaussian sigma 1.5, C1=0.01^2, C2=0.03^2 on [0,1] data)
# --------------------------------------------------------------------------------------------
def psnr(a, b, peak=2.0):
    mse = torch.mean((a.double() - b.double()) ** 2)
    return float(10 * torch.log10(peak * peak / mse)) if mse > 0 else float("inf")


def ssim(a, b, window=11, sigma=1.5):
    a = (a.double() + 1) / 2
    b = (b.double() + 1) / 2
    g = torch.exp(-((torch.arange(window, dtype=torch.float64) - window // 2) ** 2) / (2 * sigma * sigma))
    g = g / g.sum()
    C = a.shape[1]
    w = (g[:, None] * g[None, :]).expand(C, 1, window, window).contiguous()
    pad = window // 2
    mu1 = F.conv2d(a, w, padding=pad, groups=C)
    mu2 = F.conv2d(b, w, padding=pad, groups=C)
    s11 = F.conv2d(a * a, w, padding=pad, groups=C) - mu1 * mu1
    s22 = F.conv2d(b * b, w, padding=pad, groups=C) - mu2 * mu2
    s12 = F.conv2d(a * b, w, padding=pad, groups=C) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return float(m.mean())
