import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import torch
from cfen_vit_dehazing_amd import ops, packing
d = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)
def timeit(f, n=10):
    f(); f(); torch.cuda.synchronize(); tot = 0.0
    for _ in range(n):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); tot += s.elapsed_time(e)
    return tot / n * 1e3
D, H = 384, 1536
for M in (128, 8192, 24576, 32768):
    r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
    x, att = r(M, D), r(M, D)
    wp, w1, w2 = r(D, D, sc=D ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5)
    g, b = torch.ones(D, device=d), torch.zeros(D, device=d)
    b1, b2 = torch.zeros(H, device=d), torch.zeros(D, device=d)
    kd, kh = packing.kperm32(D).to(d), packing.kperm32(H).to(d)
    sa, sp = packing.pack_stream_pair(w1[:, kd], w2[:, kh]), packing.pack_stream_sq(wp)
    res = []
    for dbg in (0, 1, 2):
        ops.tune("mlp3.debug", dbg)
        res.append(round(timeit(lambda: ops.mlp_stream_block(x, sa, b1, b2, H, ln=(g, b), second=(sa, b1, b2), proj=(att, sp))), 1))
    ops.tune("mlp3.debug", 0)
    print("M=%d: full %.1f us, no DMA refill %.1f us, no MFMA %.1f us" % (M, *res), flush=True)
