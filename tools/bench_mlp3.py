#!/usr/bin/env python3
"""k_mlp3 (fragment-stream MLP block, csrc/k_stream.hip) against the launches it replaces, cold caches (a 256 MiB fill between calls):
LViT level 3 / GViT level 1 (D = 384): proj + ln2_ffn1 + ffn2 + head1 + head2 as five k_gemm_dma launches vs one k_mlp3 launch;
LViT level 2 (D = 192): k_mlp2 vs k_mlp3."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing

d = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)


def timeit(f, n=12):
    f(); f()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / n * 1e3


def run(D, M, hdr=4):
    H = hdr * D
    r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
    x, att = r(M, D), r(M, D)
    wp, w1a, w2a, w1b, w2b = r(D, D, sc=D ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5)
    g, b = torch.ones(D, device=d), torch.zeros(D, device=d)
    b1, b2 = torch.zeros(H, device=d), torch.zeros(D, device=d)
    kd, kh = packing.kperm32(D).to(d), packing.kperm32(H).to(d)
    sa, sb, sp = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]), packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]), packing.pack_stream_sq(wp)
    fl = (8.0 * D * H + 2.0 * D * D) * M
    t3 = timeit(lambda: ops.mlp_stream_block(x, sa, b1, b2, H, ln=(g, b), second=(sb, b1, b2), proj=(att, sp)))
    out = {"D": D, "M": M, "H": H, "k_mlp3_us": round(t3, 1), "k_mlp3_TF": round(fl / t3 / 1e6, 1)}
    if D == 192:
        for tm in (2, 3, 4, 22, 24):
            ops.tune("mlp3.tm192", tm)
            tt = timeit(lambda: ops.mlp_stream_block(x, sa, b1, b2, H, ln=(g, b), second=(sb, b1, b2), proj=(att, sp)))
            out["k_mlp3_tm%d_us" % tm] = round(tt, 1)
        ops.tune("mlp3.tm192", 22)
    if D in (96, 192):
        w = lambda a_, k_: a_[:, k_].contiguous()
        t2 = timeit(lambda: ops.mlp_block(x, w(w1a, kd), b1, w(w2a, kh), b2, ln=(g, b), second=(w(w1b, kd), b1, w(w2b, kh), b2), proj=(att, wp)))
        out.update({"k_mlp2_us": round(t2, 1), "k_mlp2_TF": round(fl / t2 / 1e6, 1)})
    else:
        from cfen_vit_dehazing_amd.packing import ln_folded
        lf = ln_folded(None, g, b, b1, "f", torch.float16, w1a.float())
        def chain():
            x1 = ops.gemm_nt(att, wp, residual=x)
            hid = ops.gemm_ln(x1, lf["f.wl"], lf["f.s"], lf["f.bl"], relu=True)
            x2 = ops.gemm_nt(hid, w2a, bias=b2, residual=x1)
            hid = ops.gemm_nt(x2, w1b, bias=b1, relu=True)
            return ops.gemm_nt(hid, w2b, bias=b2, residual=x2)
        tc = timeit(chain)
        out.update({"gemm_chain_us": round(tc, 1), "gemm_chain_TF": round(fl / tc / 1e6, 1)})
    print(json.dumps(out), flush=True)


for D, M in ((384, 8192), (384, 24576), (384, 2048), (384, 6144), (192, 32768), (192, 98304)):
    run(D, M)
