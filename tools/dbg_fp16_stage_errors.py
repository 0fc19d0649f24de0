#!/usr/bin/env python3
"""Where does the fp16 path lose accuracy on the reference's own init distribution?  

Runs the fp16 plan and the exact-fp32 plan on the `refinit_full512` weights (the reference's ActNorm parameters loaded, so both plans normalise with
the same tables) and prints, per top-level stage (SURVEY Appendix D names, forward order): the stage's scale (rms of the fp32 tensor), the fp16 plan's
max-abs and rms difference from the fp32 plan over the WHOLE tensor, the same relative to the stage's rms, and the fp16 plan's max-abs difference
from the reference's own samples of that stage (fixture `stage_smp/*`, 512 samples per stage).  A stage whose relative error jumps against its
inputs' is where fp16 loses bits.  Usage: dbg_fp16_stage_errors.py [fixture-name] > profiles/rNN_fp16_stage_errors_refinit.txt
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

from helpers import load_net_fixture, sample_idx, weight_mode
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
from cfen_vit_dehazing_amd import ops

name = sys.argv[1] if len(sys.argv) > 1 else "refinit_full512_nf24_hdr4"
cfg, batch, z = load_net_fixture(name)
sd = generate_state_dict(cfg, seed=0, mode=weight_mode(name))
if "actnorm_names" in z:
    for k in [str(v) for v in z["actnorm_names"]]:
        sd[k + ".weight"], sd[k + ".bias"] = torch.from_numpy(z["actnorm_w/" + k]), torch.from_numpy(z["actnorm_b/" + k])
        sd[k + ".initialized"] = torch.tensor(1)
x = synthetic_input(batch, cfg).to("cuda:0")
ops.tune("net.keep_stages", 1)      # the fused tail keeps us_conv_d01* on chip otherwise
names = [str(s) for s in z["stage_names"]]


def run(dtype):
    net = dec_ipt(cfg, compute_dtype=dtype)
    net.load_state_dict(sd, strict=True)
    net.to("cuda:0")
    outs = [o.float().cpu() for o in net(x)]
    st = {}
    for n in names:
        if n.startswith("tail_"):
            continue
        st[n] = net.stage(n).float().cpu()
    xf = st["ds_conv_e01"] if "ds_conv_e01" in st else None
    for n in list(st):
        if n.startswith("lgcat_conv_d01") and xf is not None:
            st[n] = st[n] - xf
    for nm, o in zip(("tail_r", "tail_s", "tail_d"), outs):
        st["out:" + nm] = o
    del net
    return st


s32, s16 = run("fp32"), run("fp16")
print("# %s  batch %d  (fp16 plan against the exact-fp32 plan over whole stage tensors; last column: fp16 plan against the reference's stage samples)" % (name, batch))
print("%-28s %10s %10s %10s %10s %10s %10s" % ("stage", "rms(fp32)", "max|d|", "rms(d)", "max/rms", "rmsd/rms", "vs-ref-smp"))
for n in list(s32):
    a, b = s32[n], s16[n]
    d = (a - b).double()
    rms = float(a.double().pow(2).mean().sqrt())
    ref = float("nan")
    if ("stage_smp/" + n) in z:
        smp = b.flatten()[sample_idx(n, b.numel())].numpy()
        ref = float(np.abs(smp - z["stage_smp/" + n]).max())
    print("%-28s %10.3e %10.3e %10.3e %10.3e %10.3e %10.3e" % (n, rms, float(d.abs().max()), float(d.pow(2).mean().sqrt()), float(d.abs().max()) / max(rms, 1e-30),
                                                         float(d.pow(2).mean().sqrt()) / max(rms, 1e-30), ref))
