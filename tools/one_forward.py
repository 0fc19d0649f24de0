#!/usr/bin/env python3
"""A few eager single-lane forwards (batch 8, 512x512, fp16) for rocprofv3 --pmc runs."""
import sys, os
os.environ.setdefault("CFEN_SERIAL", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
net = dec_ipt(cfg, compute_dtype="fp16"); net.load_state_dict(generate_state_dict(cfg, seed=0)); net.to("cuda:0")
x = synthetic_input(8, cfg).to("cuda:0")
for _ in range(2): net(x)
torch.cuda.synchronize()
