#!/usr/bin/env python3
"""Micro-benchmark of the fused MLP kernel: time vs hidden size separates per-stage cost from prologue/epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops

def timeit(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

d = "cuda:0"
for D, M in ((96, 131072), (192, 32768), (384, 8192)):
    x = torch.randn(M, D, device=d).half()
    g, b = torch.ones(D, device=d), torch.zeros(D, device=d)
    for H in (64, 128, 256, 4 * D):
        w1, w2 = (torch.randn(H, D, device=d) * 0.1).half(), (torch.randn(D, H, device=d) * 0.1).half()
        b1, b2 = torch.zeros(H, device=d), torch.zeros(D, device=d)
        one = timeit(lambda: ops.mlp_block(x, w1, b1, w2, b2, ln=(g, b)))
        two = timeit(lambda: ops.mlp_block(x, w1, b1, w2, b2, ln=(g, b), second=(w1, b1, w2, b2)))
        fl = 4.0 * M * D * H
        print("D=%d M=%d H=%4d: one stage %.1f us (%.0f TF/s), two stages %.1f us (%.0f TF/s)" % (D, M, H, one, fl / one / 1e6, two, 2 * fl / two / 1e6))
    t = timeit(lambda: ops.layernorm(x, g, b))
    print("   layernorm pass (read+write x): %.1f us" % t)
