#!/usr/bin/env python3
"""Run a few launches of one GEMM shape/kernel for rocprofv3 --pmc.  Usage: pmc_gemm.py M N K variant [residual]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops
M, N, K, v = (int(a) for a in sys.argv[1:5])
res = len(sys.argv) > 5
d = "cuda:0"
x = torch.randn(M, K, device=d).half(); w = (torch.randn(N, K, device=d) * 0.05).half(); b = torch.zeros(N, device=d)
r = torch.randn(M, N, device=d).half() if res else None
out = torch.empty(M, N, device=d).half()
ops.tune("gemm.kernel", v)
for _ in range(5):
    ops.gemm_nt(x, w, bias=b, residual=r, relu=not res, out=out)
torch.cuda.synchronize()
