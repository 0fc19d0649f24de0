import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops
d = "cuda:0"
def ref(qkv, nseq, S, heads):
    D = qkv.shape[1] // 3; dh = D // heads
    q, k, v = qkv.double().view(nseq, S, 3, heads, dh).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v
    return a.transpose(1, 2).reshape(nseq * S, D)
def hm(qkv, nseq, S, heads):
    dh = qkv.shape[1] // 3 // heads
    return qkv.view(nseq, S, 3, heads, dh).permute(0, 3, 2, 1, 4).contiguous().view(-1)
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)
for nseq, S, heads in ((128, 256, 8), (384, 256, 8), (96, 256, 16), (512, 256, 4)):
    g = torch.Generator().manual_seed(1)
    qkv = (torch.randn(nseq * S, 3 * heads * 24, generator=g) * 1.5).half()
    x = hm(qkv, nseq, S, heads).to(d)
    res = []
    for pair in (0, 1):
        ops.tune("attn.hm_pair", pair)
        out = ops.attention_head_major(x, nseq, S, heads)
        err = float((out.double().cpu() - ref(qkv, nseq, S, heads)).abs().max()) if nseq <= 128 else -1
        tot = 0
        for _ in range(10):
            flush.fill_(1)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.attention_head_major(x, nseq, S, heads); e.record(); torch.cuda.synchronize(); tot += s.elapsed_time(e)
        res.append((round(tot / 10 * 1e3, 1), err))
    ops.tune("attn.hm_pair", 0)
    print(nseq, S, heads, "single:", res[0], "pair:", res[1], flush=True)
