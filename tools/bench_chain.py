#!/usr/bin/env python3
"""Time the persistent GEMM chain (csrc/k_gvit.hip) on GViT-shaped phases, alone on the chip, caches cold (a 512 MiB fill between launches).
Usage: bench_chain.py [team ...]      prints one JSON line per (shape, team)"""
import sys, os, json, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing

teams = [int(a) for a in sys.argv[1:]] or [48]
for kv in filter(None, os.environ.get("CFEN_TUNE", "").split(",")):
    ops.tune(kv.split("=")[0], int(kv.split("=")[1]))
d = torch.device("cuda:0")
dt = torch.float16
flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)


def rnd(shape, scale=1.0):
    return (torch.randn(shape, device=d) * scale).to(dt)


sync = torch.zeros(8192 + (64 << 20), dtype=torch.uint8, device=d)


def time_chain(phases, M, team, fold=None, reps=7):
    ts = []
    for _ in range(reps):
        flush.fill_(1)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.gemm_chain(phases, M, team, fold, sync=sync)      # includes the 8 KB memset
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    assert int(sync[:8].view(torch.int32)[1].item()) == 0 or os.environ.get("CFEN_TUNE")
    return sorted(ts)[len(ts) // 2]


xn, yn, wn = rnd((16, 64)), torch.empty(16, 128, dtype=dt, device=d), packing.pack_stream_tiles(rnd((128, 64)))
print(json.dumps(dict(null_launch_us=round(time_chain([dict(x=xn, w_stream=wn, N=128, K=64, y=yn)], 16, 1), 1))), flush=True)
for (B, S, D, H) in [(8, 16, 1536, 6144), (8, 64, 768, 3072), (8, 256, 384, 1536)]:
    M = B * S
    x0, att = rnd((M, D)), rnd((M, D))
    x1, qkv, hid = torch.empty(M, D, dtype=dt, device=d), torch.empty(M, 3 * D, dtype=dt, device=d), torch.empty(M, H, dtype=dt, device=d)
    C = D // 16
    mapH = 4 * int(math.isqrt(S))
    sm = torch.zeros(B, mapH, mapH, C, dtype=dt, device=d)
    W = {k: packing.pack_stream_tiles(rnd(s, 1 / math.sqrt(s[1]))) for k, s in
         dict(e=(D, D), q=(3 * D, D), p=(D, D), f1=(H, D), f2=(D, H), h1=(H, D), h2=(D, H)).items()}
    bias = {k: torch.zeros(n, device=d) for k, n in dict(e=D, q=3 * D, f1=H, f2=D, h1=H, h2=D).items()}
    s = {k: torch.zeros(n, device=d) for k, n in dict(q=3 * D, f1=H).items()}
    pos = rnd((S, D))
    for team in teams:
        def nsp(N, K):
            n, units = 1, ((M + 127) // 128) * (N // 128)
            while units * n * 2 <= team and n < 8 and (K // 64) % (2 * n) == 0 and K // 64 // (2 * n) >= 4:
                n *= 2
            return n
        pa = [dict(x=x0, w_stream=W["e"], N=D, K=D, y=x1, bias=bias["e"], residual=x0, pos=pos, nsplit=nsp(D, D)),
              dict(x=x1, w_stream=W["q"], N=3 * D, K=D, y=qkv, bias=bias["q"], lnf_s=s["q"])]
        pb = [dict(x=att, w_stream=W["p"], N=D, K=D, y=x1, residual=x1, nsplit=nsp(D, D)),
              dict(x=x1, w_stream=W["f1"], N=H, K=D, y=hid, bias=bias["f1"], lnf_s=s["f1"], relu=True),
              dict(x=hid, w_stream=W["f2"], N=D, K=H, y=x1, bias=bias["f2"], residual=x1, nsplit=nsp(D, H)),
              dict(x=x1, w_stream=W["h1"], N=H, K=D, y=hid, bias=bias["h1"], relu=True),
              dict(x=hid, w_stream=W["h2"], N=D, K=H, y=sm, bias=bias["h2"], residual=x1, nsplit=nsp(D, H), fold=True)]
        ta = time_chain(pa, M, team)
        tb = time_chain(pb, M, team, fold=(mapH, mapH, C, C, 4))
        singles = [time_chain([ph], M, team, fold=(mapH, mapH, C, C, 4)) for ph in pa + pb]
        wbytes = (5 * D * D + 4 * D * H) * 2
        print(json.dumps(dict(M=M, D=D, H=H, team=team, chain_a_us=round(ta, 1), chain_b_us=round(tb, 1), singles_us=[round(t, 1) for t in singles],
                              weight_MB=round(wbytes / 1e6, 1), weight_TBps=round(wbytes / (ta + tb) / 1e6, 2))), flush=True)
