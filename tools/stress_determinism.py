#!/usr/bin/env python3
"""Repeat eager / graph forwards and report any run whose outputs (or stages) differ bitwise from the first."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input

from cfen_vit_dehazing_amd import ops
ops.tune("net.keep_stages", 1)        # the stage maps the fused tail keeps on chip by default are compared too
small = len(sys.argv) > 1 and sys.argv[1] == "tiny"
cfg = NetConfig(24, 4, patch_size=8, load_size=64) if small else NetConfig(24, 4, patch_size=32, load_size=256)
B = 2
net = dec_ipt(cfg, compute_dtype="fp16")
net.load_state_dict(generate_state_dict(cfg, seed=0))
net.to("cuda:0")
x = synthetic_input(B, cfg).to("cuda:0")
ref = [o.clone() for o in net(x)]
names = [n for n in ("head", "ds_conv_e01", "localvit_encoder_01", "globalvit_encoder_01", "lgcat_conv_e01", "ds_conv_e02",
                     "localvit_encoder_02", "globalvit_encoder_02", "lgcat_conv_e02", "ds_conv_e03", "localvit_encoder_03",
                     "globalvit_encoder_03", "lgcat_conv_e03", "localvit_decoder_03r", "globalvit_decoder_03r", "lgcat_conv_d03r",
                     "us_conv_d03r", "sk_conv_d03r", "lgcat_conv_d02r", "us_conv_d02r", "sk_conv_d02r", "lgcat_conv_d01r", "us_conv_d01r",
                     "localvit_decoder_03s", "globalvit_decoder_03s", "lgcat_conv_d03s", "us_conv_d03s", "sk_conv_d03s", "lgcat_conv_d02s",
                     "us_conv_d02s", "lgcat_conv_d01s", "us_conv_d01s", "lgcat_conv_d03d", "us_conv_d03d", "cfsm2g_d03d", "lgcat_conv_d02d",
                     "cfsm2g_d02d", "lgcat_conv_d01d", "us_conv_d01d")]
ref_st = {n: net.stage(n).clone() for n in names}
gid, gout = net.capture(x)
bad = 0
noise_on = os.environ.get("NOISE")
big = torch.randn(8192, 4096, device="cuda:0").half() if noise_on else None
side = torch.cuda.Stream()
main = torch.cuda.Stream() if os.environ.get("OWNSTREAM") else torch.cuda.current_stream()
torch.cuda.synchronize()
for it in range(int(os.environ.get("ITERS", "40"))):
    mode = os.environ.get("MODE") or ("graph" if it % 2 else "eager")
    if noise_on:
        with torch.cuda.stream(side):
            for _ in range(6): torch.mm(big, big[:4096].t())
    with torch.cuda.stream(main):
        if mode == "graph":
            for o in gout: o.zero_()
            net.replay(gid)
            outs = gout
        else:
            outs = net(x)
    torch.cuda.synchronize()
    diffs = [float((a - b).abs().max()) for a, b in zip(ref, outs)]
    if any(d != 0 for d in diffs):
        bad += 1
        first = next((n for n in names if not torch.equal(ref_st[n], net.stage(n))), None)
        print("iter %d %s: output diffs %s; first differing stage: %s" % (it, mode, diffs, first))
print("done, %d bad of %s" % (bad, os.environ.get("ITERS", "40")))
