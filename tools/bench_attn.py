#!/usr/bin/env python3
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
for nseq, S, heads, dh in ((512, 256, 4, 24), (128, 256, 8, 24), (32, 256, 16, 24), (8, 256, 4, 96), (8, 64, 8, 96)):
    qkv = torch.randn(nseq * S, 3 * heads * dh, device=d).half()
    t = timeit(lambda: ops.attention(qkv, nseq, S, heads))
    fl = 4.0 * nseq * S * S * heads * dh
    print("nseq=%d S=%d heads=%d dh=%d: %.1f us (%.0f TF/s)  dbg=%s" % (nseq, S, heads, dh, t, fl / t / 1e6, os.environ.get("CFEN_ATTN_DBG", "0")))
