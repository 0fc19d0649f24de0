"""is k_mlp3's DMA skeleton (mlp3.debug = 2: no MFMAs) faster when the weight stream is L2-resident?  (H = 256: 0.8 MB per pair, no cache flush)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing
d = "cuda:0"
flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)
def timeit(f, n=10, cold=False):
    f(); f(); torch.cuda.synchronize(); tot = 0.0
    for _ in range(n):
        if cold: flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); tot += s.elapsed_time(e)
    return tot / n * 1e3
D = 384
for H in (256, 768, 1536):
  for M in (128, 8192, 24576):
    r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
    x = r(M, D)
    w1, w2 = r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5)
    b1, b2 = torch.zeros(H, device=d), torch.zeros(D, device=d)
    kd, kh = packing.kperm32(D).to(d), packing.kperm32(H).to(d)
    sa = packing.pack_stream_pair(w1[:, kd], w2[:, kh])
    res = []
    for dbg in (0, 2):
        ops.tune("mlp3.debug", dbg)
        for cold in (False, True):
            res.append(round(timeit(lambda: ops.mlp_stream_block(x, sa, b1, b2, H, second=(sa, b1, b2)), cold=cold), 1))
    ops.tune("mlp3.debug", 0)
    nph = 4 * H // 32
    print("H=%d M=%d (%d phases): full hot %.1f cold %.1f | no-MFMA hot %.1f cold %.1f us  -> per phase %.2f / %.2f us" % (H, M, nph, *res, res[2] / nph, res[3] / nph), flush=True)
