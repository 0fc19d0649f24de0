#!/usr/bin/env python3
"""Time every token-GEMM shape of the generator (batch B, 512x512, n_feats 24) on both GEMM kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
d = "cuda:0"

def timeit(f, n=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

shapes = []
for lvl, (T, D, G_T, G_D) in enumerate(((16384, 96, 256, 384), (4096, 192, 64, 768), (1024, 384, 16, 1536)), 1):
    for kind, t, dd in (("L", T, D), ("G", G_T, G_D)):
        M = B * t
        shapes += [("%s%d embed/proj" % (kind, lvl), M, dd, dd), ("%s%d qkv" % (kind, lvl), M, 3 * dd, dd)]
        if kind == "G" or dd == 384:
            shapes += [("%s%d ffn1" % (kind, lvl), M, 4 * dd, dd), ("%s%d ffn2" % (kind, lvl), M, dd, 4 * dd)]
VARIANTS = (0, 1, 4, 14, 5, 15, 25)
NAMES = ("tiled", "skinny", "dma64", "dma64x3", "dma32", "dma32x3", "dma32x4")
print("%-16s %7s %6s %6s | " % ("shape", "M", "N", "K") + " ".join("%9s" % n for n in NAMES) + " | best")
for name, M, N, K in shapes:
    x = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) * 0.05).half(); b = torch.zeros(N, device=d)
    r = None; out = torch.empty(M, N, device=d).half()
    res = []
    for v in VARIANTS:
        ops.tune("gemm.kernel", v)
        try:
            res.append(timeit(lambda: ops.gemm_nt(x, w, bias=b, residual=r, out=out)))
        except Exception:
            res.append(float("inf"))
    ops.tune("gemm.kernel", -1)
    auto = timeit(lambda: ops.gemm_nt(x, w, bias=b, residual=r, out=out))
    fl = 2.0 * M * N * K
    print("%-16s %7d %6d %6d | " % (name, M, N, K) + " ".join("%9.1f" % r for r in res) + " | %s %.0f TF (auto %.1f us)" % (NAMES[res.index(min(res))], fl / min(res) / 1e6, auto))
