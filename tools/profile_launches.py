#!/usr/bin/env python3
"""Per-launch timing table of one forward (HIP events around every launch, serial): label, class, ms, TFLOP/s.
Usage: profile_launches.py [batch] [dtype] [load_size] [hidden_dim_ratio]      (load_size 512 = 1024x1024 images, patch_size = load_size / 8)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input

from cfen_vit_dehazing_amd import ops
for kv in filter(None, os.environ.get("CFEN_TUNE", "").split(",")):
    ops.tune(kv.split("=")[0], int(kv.split("=")[1]))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dt = sys.argv[2] if len(sys.argv) > 2 else "fp16"
ls = int(sys.argv[3]) if len(sys.argv) > 3 else 256
hdr = int(sys.argv[4]) if len(sys.argv) > 4 else 4
cfg = NetConfig(24, hdr, patch_size=ls // 8, load_size=ls)
net = dec_ipt(cfg, compute_dtype=dt); net.load_state_dict(generate_state_dict(cfg, seed=0)); net.to("cuda:0")
x = synthetic_input(B, cfg).to("cuda:0")
for _ in range(3): net(x)
torch.cuda.synchronize()
runs = [net.profile(x)["launches"] for _ in range(5)]
n = len(runs[0])
rows = []
for i in range(n):
    ms = sorted(r[i][3] for r in runs)[2]
    rows.append((runs[0][i][0], runs[0][i][1], runs[0][i][2], ms, runs[0][i][4]))
tot = sum(r[3] for r in rows)
print("%d launches, sum %.3f ms (batch %d, %s, %dx%d, hidden_dim_ratio %d)" % (n, tot, B, dt, cfg.image_size, cfg.image_size, hdr))
for lab, cls, fl, ms, kn in rows:
    print("%-42s %-9s %8.1f us %8s  %s" % (lab, cls, ms * 1e3, ("%.0f TF" % (fl / ms / 1e9)) if fl else "", kn))
# aggregate by step kind
agg = {}
for lab, cls, fl, ms, kn in rows:
    key = (lab.split(":")[1] if ":" in lab else "conv") + (" L%s" % lab.split(":")[0][-2 if lab.split(":")[0][-1] in "rsd" else -1] if ":" in lab and "vit" in lab else "")
    key = ("G " if lab.startswith("global") else "L " if lab.startswith("local") else "") + key
    a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
print("---- by step")
for k, (c, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-28s x%-3d %8.1f us  %5.1f%%  %s" % (k, c, ms * 1e3, 100 * ms / tot, ("%.0f TF" % (fl / ms / 1e9)) if fl else ""))
print("---- by device kernel")
byk = {}
for lab, cls, fl, ms, kn in rows:
    a = byk.setdefault(kn, [0, 0.0, 0.0]); a[0] += 1; a[1] += ms; a[2] += fl
for k, (c, ms, fl) in sorted(byk.items(), key=lambda kv: -kv[1][1]):
    print("%-60s x%-3d %8.1f us  %5.1f%%  %s" % (k, c, ms * 1e3, 100 * ms / tot, ("%.0f TF" % (fl / ms / 1e9)) if fl else ""))
