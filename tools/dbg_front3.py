import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import torch, cfen_oracle
from cfen_vit_dehazing_amd import ops, packing
d = "cuda:0"
def rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)
dtype = torch.float16
for (H, W, ws, hm) in ((64, 32, 32, True), (32, 64, 16, False), (64, 32, 32, False), (32, 64, 16, True)):
    B, C, p, D, heads = 2, 96, 2, 384, 16
    S = (ws // p) ** 2
    fmap = ops.to_nhwc(rnd((B, C, H, W), 1, dtype)).to(d)
    we, be = rnd((D, D), 2, dtype, D ** -0.5), 0.1 * rnd((D,), 3, torch.float32)
    pos = rnd((S, D), 4, dtype)
    g, b = 1 + 0.1 * rnd((D,), 5, torch.float32), 0.1 * rnd((D,), 6, torch.float32)
    wq = rnd((3 * D, D), 7, dtype, D ** -0.5)
    tok = ops.patchify(fmap, C, ws, p).double().cpu()
    M = tok.shape[0]
    y = tok @ we.double().t() + be.double() + tok + pos.double().repeat(M // S, 1)
    qkv = cfen_oracle.layer_norm(y, g.double(), b.double()) @ wq.double().t()
    kd = packing.kperm32(D)
    x1, got = ops.embed_qkv(fmap, C, ws, p, packing.pack_stream_rows(we[:, kd]).to(d), be.to(d), pos.to(d), g.to(d), b.to(d),
                            packing.pack_stream_rows(wq[:, kd]).to(d), head_major_heads=heads if hm else 0, stream_weights=True)
    if hm:
        want = qkv.view(M // S, S, 3, heads, 24).permute(0, 3, 2, 1, 4).contiguous().view(-1)
        diff = (got.view(-1).double().cpu() - want).abs()
        print(H, W, ws, hm, "max", float(diff.max()))
    else:
        diff = (got.double().cpu() - qkv).abs()
        bad = (diff > 0.05)
        print(H, W, ws, hm, "max", float(diff.max()), "bad elems", int(bad.sum()), "of", diff.numel())
        if bad.any():
            rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
            print("  bad rows", rows[:20].tolist(), "n", len(rows)); print("  bad col tiles", sorted(set((cols // 16).tolist()))[:40])
