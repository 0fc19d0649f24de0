import sys, os, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import cfen_oracle
from cfen_vit_dehazing_amd import ops, packing
d = "cuda:0"
def rnd(shape, seed, dtype, sc=1.0):
    g = torch.Generator().manual_seed(seed); return (torch.randn(shape, generator=g) * sc).to(dtype)
D, H, M = 192, 768, 512
dtype = torch.float16
x, att = rnd((M, D), 1, dtype), rnd((M, D), 2, dtype)
wp = rnd((D, D), 3, dtype, D ** -0.5)
g, b = 1 + 0.1 * rnd((D,), 4, torch.float32), 0.1 * rnd((D,), 5, torch.float32)
w1a, w2a = rnd((H, D), 6, dtype, D ** -0.5), rnd((D, H), 7, dtype, 0.5 * H ** -0.5)
w1b, w2b = rnd((H, D), 8, dtype, D ** -0.5), rnd((D, H), 9, dtype, 0.5 * H ** -0.5)
b1a, b2a, b1b, b2b = (0.1 * rnd((n,), 10 + i, torch.float32) for i, n in enumerate((H, D, H, D)))
kd, kh = packing.kperm32(D), packing.kperm32(H)
sa = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]).to(d); sb = packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]).to(d); sp = packing.pack_stream_sq(wp).to(d)
xd = x.double()
ffn = lambda v, w1, b1, w2, b2, ln: v + torch.relu((cfen_oracle.layer_norm(v, g.double(), b.double()) if ln else v) @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
full = ffn(ffn(xd + att.double() @ wp.double().t(), w1a, b1a, w2a, b2a, True), w1b, b1b, w2b, b2b, False)
res = {}
for tm in (22, 24, 2, 3, 4):
    ops.tune("mlp3.tm192", tm)
    res[tm] = {}
    res[tm]["full"] = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)), proj=(att.to(d), sp)).float().cpu()
    res[tm]["noln"] = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H).float().cpu()
    res[tm]["ln"] = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d))).float().cpu()
for tm in res:
    print(tm, "vs fp64 %.3e" % float((res[tm]["full"].double() - full).abs().max()), {k: "%.3e (%d elems)" % (float((res[tm][k] - res[22][k]).abs().max()), int((res[tm][k] != res[22][k]).sum())) for k in res[tm]})
