#!/usr/bin/env python3
"""GPU: outputs with the tile-major GViT weight layout (default) next to the row-major one (CFEN_WTILE=0).  Not bitwise: with tile-major
weights the <= 128-token GEMMs run on k_gemm_dma instead of k_gemm_skinny (another order of the K sum); differences are rounding noise
(measured 9e-4 fp16, 9e-7 fp32 on the outputs)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input

for cfg, B, dt in ((NetConfig(24, 4, patch_size=8, load_size=64), 2, "fp16"), (NetConfig(24, 4, patch_size=8, load_size=64), 2, "fp32"),
                   (NetConfig(24, 4, patch_size=32, load_size=256), 2, "fp16")):
    sd = generate_state_dict(cfg, seed=0)
    x = synthetic_input(B, cfg).to("cuda:0")
    outs = []
    for wt in (False, True):
        net = dec_ipt(cfg, compute_dtype=dt)
        net.wtile = wt
        net.load_state_dict(sd, strict=True)
        net.to("cuda:0")
        outs.append([o.clone() for o in net(x)])
    worst = max(float((a - b).abs().max()) for a, b in zip(*outs))
    print(cfg.load_size, dt, "max |row-major - tile-major| = %.3e" % worst, "bitwise" if all(torch.equal(a, b) for a, b in zip(*outs)) else "DIFFERENT")
