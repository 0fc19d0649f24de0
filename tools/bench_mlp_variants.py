#!/usr/bin/env python3
"""A/B of the fused token-MLP kernel variants (cfen_tune "mlp.small_tiles") at the LViT shapes of the batch-8 512x512 forward,
interleaved rounds in one process; every variant's output is compared with variant 3 (the round-1 kernel)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops
from cfen_vit_dehazing_amd.packing import kperm32

d = "cuda:0"
flush = torch.empty(320 * 1024 * 1024, dtype=torch.uint8, device=d)


def timed(f):
    flush.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


for D, M, variants in ((96, 131072, (3, 10, 20, 30, 40)), (96, 3 * 131072, (3, 10, 20, 30, 40)), (192, 32768, (3, 10, 11, 12)), (192, 3 * 32768, (3, 10, 11, 12))):
    H = 4 * D
    g = torch.Generator(device="cpu").manual_seed(D)
    x = torch.randn(M, D, generator=g).half().to(d)
    att = torch.randn(M, D, generator=g).half().to(d)
    wp = (torch.randn(D, D, generator=g) * D ** -0.5).half().to(d)
    lg, lb = (1 + 0.1 * torch.randn(D, generator=g)).to(d), (0.1 * torch.randn(D, generator=g)).to(d)
    ws = []
    for _ in range(2):
        w1 = (torch.randn(H, D, generator=g) * D ** -0.5)[:, kperm32(D)].contiguous().half().to(d)
        w2 = (torch.randn(D, H, generator=g) * H ** -0.5)[:, kperm32(H)].contiguous().half().to(d)
        ws += [w1, (0.1 * torch.randn(H, generator=g)).to(d), w2, (0.1 * torch.randn(D, generator=g)).to(d)]
    run = lambda: ops.mlp_block(x, *ws[:4], ln=(lg, lb), second=tuple(ws[4:]), proj=(att, wp))
    ref, times = None, {v: [] for v in variants}
    for v in variants:
        ops.tune("mlp.small_tiles", v)
        y = run()
        torch.cuda.synchronize()
        if ref is None:
            ref = y.float()
        else:
            print("   D=%d M=%d variant %d vs variant %d: max-abs %.3e" % (D, M, v, variants[0], float((y.float() - ref).abs().max())))
    for rnd in range(7):
        for v in variants:
            ops.tune("mlp.small_tiles", v)
            times[v].append(timed(run))
    fl = (8.0 * H + 2.0 * D) * M * D
    for v in variants:
        t = sorted(times[v])
        print("D=%d M=%d variant %2d: median %.1f us (min %.1f)  %.0f TF/s" % (D, M, v, t[len(t) // 2], t[0], fl / t[len(t) // 2] / 1e6))
ops.tune("mlp.small_tiles", 10)
