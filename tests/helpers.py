"""Shared helpers for the parity tests (fixture loading; no reference import: /root/reference does
not exist on the GPU box)."""
import os
import zlib

import numpy as np
import torch

from cfen_vit_dehazing_amd.config import NetConfig

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_net_fixture(name):
    z = np.load(os.path.join(GOLDEN, "net_%s.npz" % name))
    nf, hdr, ps, ls = [int(v) for v in z["cfg"]]
    cfg = NetConfig(nf, hdr, patch_size=ps, load_size=ls, variant=variant_of(name))
    return cfg, int(z["batch"]), z


def variant_of(name):
    """fixture name -> generator variant (config.VARIANTS): [refinit_]{cfs,crs,v5}_* are the sibling generators, the rest is v3"""
    base = name[len("refinit_"):] if name.startswith("refinit_") else name
    for v in ("cfs", "crs", "v5"):
        if base.startswith(v + "_"):
            return v
    return "v3"


def weight_mode(name):
    """fixtures named refinit_* were made with the reference's own init distribution and uninitialised ActNorm (tools/gen_golden.py)"""
    return "reference_init" if name.startswith("refinit") else "trained"


def sample_idx(name, numel, n=512):
    g = torch.Generator()
    g.manual_seed(zlib.crc32(name.encode()) & 0x7FFFFFFF)
    return torch.randint(0, numel, (n,), generator=g)


def check_stages(z, stages, tol, rel_sum=1e-4):
    """Compare a dict of stage tensors (NCHW, CPU) against the fixture's samples and checksums."""
    worst = 0.0
    for n in [str(s) for s in z["stage_names"]]:
        t = stages[n].float().cpu()
        assert tuple(t.shape) == tuple(int(v) for v in z["stage_shape/" + n]), n
        smp = t.flatten()[sample_idx(n, t.numel())].numpy()
        d = float(np.abs(smp - z["stage_smp/" + n]).max())
        worst = max(worst, d)
        assert d <= tol, "stage %s: sampled max-abs %.3e > %.1e" % (n, d, tol)
        ref_abs = float(z["stage_abs/" + n])
        got_abs = float(t.double().abs().sum())
        assert abs(got_abs - ref_abs) <= rel_sum * ref_abs + 1e-6, "stage %s abs-sum %.6f vs %.6f" % (n, got_abs, ref_abs)
    return worst


def check_outputs(z, outs, tol):
    worst = 0.0
    for nm, o in zip(("xr", "xs", "xd"), outs):
        o = o.float().cpu()
        if ("out/" + nm) in z:
            d = float((o - torch.from_numpy(z["out/" + nm])).abs().max())
        else:
            n = o.shape[-1]
            c0 = n // 2 - 32
            d2 = float((o[:, :, 3::8, 5::8] - torch.from_numpy(z["strided/" + nm])).abs().max())
            d1 = float((o[:, :, c0:c0 + 64, c0:c0 + 64] - torch.from_numpy(z["crop/" + nm])).abs().max()) if ("crop/" + nm) in z else d2
            d = max(d1, d2)
        worst = max(worst, d)
        assert d <= tol, "%s: max-abs %.3e > %.1e" % (nm, d, tol)
    return worst
