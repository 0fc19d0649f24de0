#!/usr/bin/env python3
"""Repeat the forward and count, per named stage, the runs whose stage map differs bitwise from the first run."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
net = dec_ipt(cfg, compute_dtype="fp16"); net.load_state_dict(generate_state_dict(cfg, seed=0)); net.to("cuda:0")
x = synthetic_input(2, cfg).to("cuda:0")
stages = sys.argv[1:]
net(x); torch.cuda.synchronize()
ref = {s: net.stage(s).clone() for s in stages}
bad = {s: 0 for s in stages}
N = 16
for it in range(N):
    net(x); torch.cuda.synchronize()
    for s in stages:
        bad[s] += not torch.equal(net.stage(s), ref[s])
print({s: "%d/%d runs differ" % (bad[s], N) for s in stages})
