#!/usr/bin/env python3
"""Token-GEMM kernel variants with COLD caches (a 256 MiB fill between launches evicts L2 and the Infinity Cache, as the
540 MB of weights streaming through the network do).  Run under rocprofv3 --kernel-trace; prints nothing itself:
  rocprofv3 --kernel-trace --output-format csv -d out -- python3 tools/bench_gemm_cold.py ; python3 tools/bench_gemm_cold.py --report out/*/*kernel_trace.csv"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SHAPES = [("G3 emb", 128, 1536, 1536), ("G3 qkv", 128, 4608, 1536), ("G3 ffn1", 128, 6144, 1536), ("G3 ffn2", 128, 1536, 6144),
          ("G2 emb", 512, 768, 768), ("G2 qkv", 512, 2304, 768), ("G2 ffn1", 512, 3072, 768), ("G2 ffn2", 512, 768, 3072),
          ("G1 emb", 2048, 384, 384), ("G1 ffn1", 2048, 1536, 384), ("G1 ffn2", 2048, 384, 1536),
          ("L3 emb", 8192, 384, 384), ("L3 ffn1", 8192, 1536, 384), ("L3 ffn2", 8192, 384, 1536), ("L2 qkv", 32768, 576, 192)]
VARIANTS = (1, 4, 14, 5, 15, 25)
NAMES = ("skinny", "dma64", "dma64x3", "dma32", "dma32x3", "dma32x4")
REPS = 8
if len(sys.argv) > 2 and sys.argv[1] == "--report":
    import csv, statistics
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    g = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if "gemm" in r["Kernel_Name"]]
    print("%-8s " % "" + " ".join("%8s" % n for n in NAMES))
    for i, (n, M, N, K) in enumerate(SHAPES):
        print("%-8s " % n + " ".join("%8.1f" % statistics.median(g[(i * len(VARIANTS) + v) * REPS:(i * len(VARIANTS) + v + 1) * REPS]) for v in range(len(VARIANTS))))
    sys.exit(0)
import torch
from cfen_vit_dehazing_amd import ops
d = "cuda:0"
junk = torch.empty(256 * 1024 * 1024, dtype=torch.uint8, device=d)
for name, M, N, K in SHAPES:
    x = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) * 0.05).half(); out = torch.empty(M, N, device=d).half()
    for v in VARIANTS:
        ops.tune("gemm.kernel", v)
        for _ in range(REPS):
            junk.fill_(1)
            ops.gemm_nt(x, w, out=out)
    torch.cuda.synchronize()
