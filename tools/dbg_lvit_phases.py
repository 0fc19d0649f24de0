"""share of the MLP phase in k_lvit_window: full hidden width (384) against 32 hidden units (one chunk per MLP stage instead of twelve)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.manifest import generate_state_dict
d = "cuda:0"
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
g = cfg.vit("localvit_encoder_01")
sd = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in generate_state_dict(cfg, seed=0, with_dead=False).items() if k.startswith(g.name + ".")}
pk = packing.pack_vit(sd, g, torch.float16)
pk.update(packing.pack_lvit_window(sd, g, torch.float16))
pk = {k: v.to(d).contiguous() for k, v in pk.items()}
n = g.name
small = dict(pk)
# hidden 32: the stream's first MLP chunk of either pair (fragments of 512 fp16 elements: 102 before the MLP chunks, 12 per chunk, 12 chunks per pair)
ws = pk[n + ".lw.ws"]
pre, chunk = 102 * 512, 12 * 512
small[n + ".lw.ws"] = torch.cat((ws[:pre + chunk], ws[pre + 12 * chunk:pre + 13 * chunk])).contiguous()
for a in ("ffn1", "head1"):
    small[n + "." + a + ".b"] = pk[n + "." + a + ".b"][:32].contiguous()
flush = torch.empty(320 << 20, dtype=torch.uint8, device=d)
def timed(f, n=9):
    ts = []
    for _ in range(n):
        flush.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]
for B in (8, 24):
    x = (torch.rand(B, 256, 256, 24, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().to(d)
    for shape in (0, 1):
        ops.tune("lvit.shape", shape)
        full = timed(lambda: ops.lvit_window(x, 24, 32, 2, pk, n, 384))
        one = timed(lambda: ops.lvit_window(x, 24, 32, 2, small, n, 32))
        print("B=%d shape %d: hidden 384 %.1f us, hidden 32 %.1f us -> 22 more MLP chunks cost %.1f us (%.2f us per chunk and round of workgroups)"
              % (B, shape, full, one, full - one, (full - one) / 22 / (B * 64 / 256)), flush=True)
ops.tune("lvit.shape", 0)
