#!/usr/bin/env python3
"""k_lvit_window<6, 16, 1>: where one window's time goes -- s_memtime at the section boundaries of workgroup 0 ("lvit.debug" = 64, printed by the library on stderr), for 512 and 1536
windows (the encoder launch and the grouped decoder launch at batch 8), and the launch times.  Usage: dbg_lvit_sections.py [lvit.shape ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.manifest import generate_state_dict

d = "cuda:0"
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
g = [v for v in cfg.vit_instances() if v.name == "localvit_encoder_01"][0]
sd = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in generate_state_dict(cfg, seed=0, with_dead=False).items() if k.startswith(g.name + ".")}
pk = packing.pack_vit(sd, g, torch.float16)
pk.update(packing.pack_lvit_window(sd, g, torch.float16))
pk = {k: v.to(d).contiguous() for k, v in pk.items()}
flush = torch.empty(320 << 20, dtype=torch.uint8, device=d)


def timed(f, n=9):
    ts = []
    for _ in range(n):
        flush.zero_()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e) * 1e3)
    return sorted(ts)[n // 2]


shapes = [int(v) for v in sys.argv[1:]] or [2]
for B in (8, 24):
    x = (torch.rand(B, 256, 256, 24, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().to(d)
    for shape in shapes:
        ops.tune("lvit.shape", shape)
        ops.tune("lvit.debug", 0)
        call = lambda: ops.lvit_window(x, 24, 32, 2, pk, g.name, 384)
        call(); torch.cuda.synchronize()
        print("B=%d (%d windows) lvit.shape %d: %.1f us" % (B, B * 64, shape, timed(call)), flush=True)
        sys.stderr.write("==== B=%d lvit.shape %d\n" % (B, shape)); sys.stderr.flush()
        ops.tune("lvit.debug", 64)
        for _ in range(2):
            flush.zero_(); call(); torch.cuda.synchronize()
        ops.tune("lvit.debug", 0)
ops.tune("lvit.shape", 2)
