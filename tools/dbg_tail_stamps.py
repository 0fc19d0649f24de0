#!/usr/bin/env python3
"""Per-step stamps of k_tail_fused's three wave groups (tail.debug = 64: workgroup 0 writes s_memrealtime after each barrier and when its own work is done;
the library prints them to stderr after the launch).  Usage: dbg_tail_stamps.py [segments]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
from cfen_vit_dehazing_amd import ops
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
net = dec_ipt(cfg, compute_dtype="fp16"); net.load_state_dict(generate_state_dict(cfg, seed=0)); net.to("cuda:0")
x = synthetic_input(8, cfg).to("cuda:0")
ops.tune("tail.segments", int(sys.argv[1]) if len(sys.argv) > 1 else 4)
for _ in range(2): net(x)
torch.cuda.synchronize()
sys.stderr.write("==== stamped forward\n")
ops.tune("tail.debug", 64)
net(x)
torch.cuda.synchronize()
ops.tune("tail.debug", 0)
