#!/usr/bin/env python3
"""Evidence for / against the round-1 claim "packed-fp32 VALU instructions return wrong high-lane values beside another kernel's MFMA
waves" (cfen_vit_dehazing_amd/build.py DEVICE_FLAGS).  Two experiments on the MI355X box, results as one JSON object:

 1. tools/repro/pk_fma_repro.hip -- a stand-alone kernel pair with no shared buffers: v_pk_fma_f32 / v_pk_mul_f32 chains checked
    against scalar twins, alone and beside an MFMA-spinning kernel on a second stream.
 2. The library itself built twice from the SAME sources -- shipped flags (packed-fp32 off) and with packed-fp32 code generation ON
    (into /tmp) -- each running the bitwise-reproducibility stress of the two-lane forward (eager and hipGraph, with GEMM noise on a
    third stream) in its own process.  This separates the compiler flag from the cross-lane-reduction rewrite that landed in the same
    round-1 commit: the reductions are in both builds.

    python tools/repro_pk_fma.py [--iters 40] > profiles/r02_pk_fma_repro.json
"""
import argparse, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STRESS = r'''
import os, sys, json
sys.path.insert(0, %(root)r)
import torch
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.hipnet import dec_ipt
from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
net = dec_ipt(cfg, compute_dtype="fp16"); net.load_state_dict(generate_state_dict(cfg, seed=0)); net.to("cuda:0")
x = synthetic_input(4, cfg).to("cuda:0")
ref = [o.clone() for o in net(x)]
gid, gout = net.capture(x)
big = torch.randn(4096, 4096, device="cuda:0").half(); side = torch.cuda.Stream()
bad = {"eager": 0, "graph": 0}; worst = 0.0
for it in range(%(iters)d):
    mode = "graph" if it %% 2 else "eager"
    with torch.cuda.stream(side):
        for _ in range(4): torch.mm(big, big)
    if mode == "graph":
        for o in gout: o.zero_()
        net.replay(gid); outs = gout
    else:
        outs = net(x)
    torch.cuda.synchronize()
    d = max(float((a - b).abs().max()) for a, b in zip(ref, outs))
    if d != 0.0: bad[mode] += 1; worst = max(worst, d)
print("STRESS " + json.dumps({"differing_runs": bad, "runs": %(iters)d, "worst_abs_diff": worst}))
'''


def run_stress(lib, iters):
    env = dict(os.environ, CFEN_HIP_LIB=lib)
    out = subprocess.run([sys.executable, "-c", STRESS % {"root": ROOT, "iters": iters}], env=env, capture_output=True, text=True, timeout=1200)
    for line in out.stdout.splitlines():
        if line.startswith("STRESS "):
            return json.loads(line[7:])
    return {"error": (out.stderr or out.stdout)[-600:]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    args = ap.parse_args()
    res = {}
    exe = os.path.join(tempfile.gettempdir(), "pk_fma_repro")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-o", exe, os.path.join(ROOT, "tools", "repro", "pk_fma_repro.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    try:
        res["standalone_kernel_pair"] = json.loads(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        res["standalone_kernel_pair"] = {"error": (out.stdout + out.stderr)[-400:]}
    from cfen_vit_dehazing_amd import build as b
    shipped = b.build()
    res["library_shipped_flags_packed_fp32_off"] = dict(run_stress(shipped, args.iters), packed_fp32_instructions=b.check_no_packed_fp32(shipped))
    alt_dir = tempfile.mkdtemp(prefix="cfen_pk_")
    alt = b.build(force=True, packed_fp32=True, lib=os.path.join(alt_dir, "libcfen_hip_pk.so"), objdir=os.path.join(alt_dir, "obj"))
    res["library_packed_fp32_on"] = dict(run_stress(alt, args.iters), packed_fp32_instructions=b.check_no_packed_fp32(alt))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
