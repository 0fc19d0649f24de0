#!/usr/bin/env python3
"""Soak of what bench.py times: N forwards in flight on replica launch plans (serial plan each, own workspace / slab, shared weights), N different
input batches; every `check` steps all streams are drained and every slab is compared BIT FOR BIT with the one-at-a-time forward of its batch.
    GPU_MAX_HW_QUEUES=8 python3 tools/stress_in_flight.py [steps=2000] [lanes=4] [check=50] [dtype=fp16] [load_size=256] [batch=8]
Prints one JSON line; exit code 1 on any mismatch."""
import json
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4
check = int(sys.argv[3]) if len(sys.argv) > 3 else 50
dtype = sys.argv[4] if len(sys.argv) > 4 else "fp16"
load = int(sys.argv[5]) if len(sys.argv) > 5 else 256
cfg = NetConfig(24, 4, patch_size=load // 8, load_size=load)
dev = torch.device("cuda", 0)
net = dec_ipt(cfg, compute_dtype=dtype)
net.load_state_dict(generate_state_dict(cfg, seed=0), strict=True)
net.to(dev)
net.serial_plan = True
B, n = (int(sys.argv[6]) if len(sys.argv) > 6 else 8), cfg.image_size
xs = [synthetic_input(B, cfg, seed0=B * k).to(dev) for k in range(N)]
want = [torch.cat([o.reshape(-1) for o in net(x)]).clone() for x in xs]
slabs = [torch.empty(7 * B * n * n, dtype=torch.float32, device=dev) for _ in range(N)]
gids = []
for k in range(N):
    net.replica = k
    gids.append(net.capture(xs[k], out=slabs[k])[0])
net.replica = 0
streams = [torch.cuda.Stream(dev) for _ in range(N)]
torch.cuda.synchronize()
bad, checks, written = 0, 0, set()
for i in range(steps):
    with torch.cuda.stream(streams[i % N]):
        net.replay(gids[i % N])
    written.add(i % N)
    if (i + 1) % check == 0 or i == steps - 1:
        torch.cuda.synchronize()
        for k in sorted(written):                 # slabs a step wrote since the last check
            bad += int(not torch.equal(slabs[k], want[k]))
            slabs[k].fill_(float("nan"))          # the next round must really rewrite it
            checks += 1
        written.clear()
        torch.cuda.synchronize()
print(json.dumps({"steps": steps, "lanes": N, "checks": checks, "slab_mismatches": bad, "dtype": dtype, "image": n, "batch": B}))
sys.exit(1 if bad else 0)
