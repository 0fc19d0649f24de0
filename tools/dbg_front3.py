#!/usr/bin/env python3
"""k_front3<24, 2, 3> (LViT level 3 / GViT level 1 front half: patch gather -> linear_encoding + residual + position -> X1, LN1 -> qkv): launch times of the shipped kernel and of the
`front3.debug` timing variants on 16 / 64 / 192 workgroups (cold caches), and the stamped sections of workgroup 0 (stderr).  Usage: dbg_front3.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing

d = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)


def timeit(f, n=12):
    f(); f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    C, p, ws, D = 96, 2, 32, 384
    torch.manual_seed(0)
    for B in (2, 8, 24):                     # 64 x 64 maps: 4 windows of 256 tokens each -> 1024 tokens per image
        H = W = 64
        M = B * (H // p) * (W // p)
        fmap = (torch.randn(B, H, W, C, device=d)).half()
        r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
        we, wq = r(D, D, sc=D ** -0.5), r(3 * D, D, sc=D ** -0.5)
        be, pos = torch.randn(D, device=d) * 0.1, r((ws // p) ** 2, D)
        g, b = torch.ones(D, device=d), torch.zeros(D, device=d)
        kd = packing.kperm32(D).to(d)
        se, sq = packing.pack_stream_rows(we[:, kd].contiguous()), packing.pack_stream_rows(wq[:, kd].contiguous())
        call = lambda: ops.embed_qkv(fmap, C, ws, p, se, be, pos, g, b, sq, head_major_heads=16, stream_weights=True)
        fl = 8.0 * M * D * D
        ops.tune("front3.debug", 0)
        med, best = timeit(call)
        print("M=%d (%d workgroups): shipped %.1f us median / %.1f best = %.0f TF" % (M, M // 128, med, best, fl / med / 1e6), flush=True)
        for v, nm in ((1, "no refills"), (8, "no qkv stores"), (24, "no stores at all")):
            ops.tune("front3.debug", v)
            med, best = timeit(call)
            print("  debug=%-3d %-20s %.1f us median / %.1f best" % (v, nm, med, best), flush=True)
        for v in (64, 64 + 1, 64 + 8, 64 + 24):
            sys.stderr.write("==== M=%d stamped, debug=%d\n" % (M, v)); sys.stderr.flush()
            ops.tune("front3.debug", v)
            for _ in range(2):
                flush.fill_(1); call(); torch.cuda.synchronize()
        ops.tune("front3.debug", 0)


if __name__ == "__main__":
    main()
