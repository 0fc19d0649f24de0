#!/usr/bin/env python3
"""Deformable-conv forward (DCNv1 / DCNv2) at the feature-map shapes of the v3 generator:
(B,24,256,256), (B,48,128,128), (B,96,64,64), 3x3, stride 1, pad 1, deformable groups 1 and 8.  Prints one JSON line per case:
kernel time (HIP events on the launch stream, median of repetitions), achieved GB/s against the ALGORITHMIC bytes of SURVEY 8d
(input + offsets (+ mask) + output + weights, each once) and the HBM peak, and the MFMA rate of its 2*Cout*C*9*H*W contraction.

    python tools/bench_dcn.py [--batch 8] [--dtype fp16] [--check]      (--check: compare one image with the C oracle)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import dcn, ops
for kv in filter(None, os.environ.get("CFEN_TUNE", "").split(",")):
    ops.tune(kv.split("=")[0], int(kv.split("=")[1]))

HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--backward", action="store_true", help="time the backward of each case (all gradients) instead of the forward")
    ap.add_argument("--burst", type=int, default=0, help="also time N back-to-back calls through the C ABI (preallocated output and scratch, no cache eviction, "
                    "no host synchronisation in between): the rate with the GPU kept busy, reported as us_burst")
    ap.add_argument("--no-flush", action="store_true", help="do not evict the caches between repetitions (hot numbers)")
    ap.add_argument("--channels-last", action="store_true", help="hand the input over as a channels_last tensor (round 6: no NHWC pre-pass, cfen_*_forward_nhwc)")
    ap.add_argument("--channels", type=int, default=0, help="only the case with this many channels (24, 48 or 96); 0 = all three")
    args = ap.parse_args()
    dt = torch.float16 if args.dtype == "fp16" else torch.float32
    esz = 2 if dt == torch.float16 else 4
    dev = "cuda:0"
    B = args.batch
    if args.check:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import dcn_oracle
    flush = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=dev)       # > Infinity Cache: every timed launch starts cold
    for C, H in ((24, 256), (48, 128), (96, 64)):
        if args.channels and C != args.channels:
            continue
        for dg in (1, 8):
            g = torch.Generator().manual_seed(C * 10 + dg)
            x = torch.randn(B, C, H, H, generator=g).to(dt).to(dev)
            if args.channels_last and not args.backward:
                x = x.contiguous(memory_format=torch.channels_last)
            w = (torch.randn(C, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(dt).to(dev)
            off = (torch.randn(B, dg * 18, H, H, generator=g) * 2.0).to(dt).to(dev)
            mask = torch.sigmoid(torch.randn(B, dg * 9, H, H, generator=g)).to(dt).to(dev)
            bias = torch.randn(C, generator=g).to(dt).to(dev)
            for ver in (1, 2):
                fn = (lambda: dcn.deform_conv(x, off, w, 1, 1, 1, 1, dg)) if ver == 1 else \
                     (lambda: dcn.modulated_deform_conv(x, off, mask, w, bias, 1, 1, 1, 1, dg))
                if args.backward:
                    leaves = [t.detach().clone().requires_grad_() for t in ((x, off, w) if ver == 1 else (x, off, mask, w, bias))]
                    y = dcn.deform_conv(*leaves, 1, 1, 1, 1, dg) if ver == 1 else dcn.modulated_deform_conv(*leaves, 1, 1, 1, 1, dg)
                    gy = torch.randn(y.shape, generator=g).to(dt).to(dev)
                    fn = lambda: torch.autograd.grad(y, leaves, gy, retain_graph=True)
                out = fn()
                torch.cuda.synchronize()
                times = []
                for _ in range(args.reps):
                    if not args.no_flush:
                        flush.zero_()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(); fn(); b.record()
                    torch.cuda.synchronize()
                    times.append(a.elapsed_time(b))
                times.sort()
                ms = times[len(times) // 2]
                px = B * H * H
                elems = C * px + (2 if ver == 1 else 3) * dg * 9 * px + C * px + C * C * 9
                if args.backward:       # reads input, offsets (+ mask), weights, grad_out; writes grad_input, grad_offset (+ grad_mask), grad_weight
                    elems = 2 * (C * px + (2 if ver == 1 else 3) * dg * 9 * px + C * C * 9) + C * px
                gbs = elems * esz / ms / 1e6
                tf = (2.0 * C * C * 9 + 8.0 * C * 9) * px / ms / 1e9 * (2 if args.backward else 1)
                rec = {"op": "dcn_v%d%s%s" % (ver, "_backward" if args.backward else "", "_channels_last_input" if args.channels_last and not args.backward else ""), "shape": [B, C, H, H], "deformable_groups": dg, "dtype": args.dtype, "us": round(ms * 1e3, 1),
                       "algorithmic_MB": round(elems * esz / 1e6, 2), "GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK, 4), "TFLOPs": round(tf, 2)}
                if args.burst and not args.backward:
                    from cfen_vit_dehazing_amd import _lib
                    from cfen_vit_dehazing_amd._lib import check, ptr, dtype_code, current_stream
                    lib = _lib.load()
                    nb = int(lib.cfen_deform_conv_columns_bytes(dtype_code(dt), B, C, H, H, C, 3, 3, 1))
                    col = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
                    o2 = torch.empty_like(out)
                    st = current_stream()
                    def call():
                        if ver == 1:
                            check(lib.cfen_deform_conv_forward(dtype_code(dt), ptr(x), ptr(w), ptr(off), ptr(o2), B, C, H, H, C, 3, 3, 1, 1, 1, 1, 1, 1, 1, dg, B,
                                                               ptr(col), nb, st), "deform_conv_forward")
                        else:
                            check(lib.cfen_modulated_deform_conv_forward(dtype_code(dt), ptr(x), ptr(w), ptr(bias), ptr(off), ptr(mask), ptr(o2), B, C, H, H, C, 3, 3,
                                                                         1, 1, 1, 1, 1, 1, 1, dg, 1, ptr(col), nb, st), "modulated_deform_conv_forward")
                    for _ in range(10):
                        call()
                    torch.cuda.synchronize()
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    for _ in range(args.burst):
                        call()
                    b.record()
                    torch.cuda.synchronize()
                    rec["us_burst"] = round(a.elapsed_time(b) * 1e3 / args.burst, 1)
                    rec["burst_equal"] = bool(torch.equal(o2, out))
                if args.check and not args.backward:
                    want = dcn_oracle.deform_conv(x[:1].float().cpu(), off[:1].float().cpu(), w.float().cpu(), 1, 1, 1, 1, dg,
                                                  **({} if ver == 1 else {"mask": mask[:1].float().cpu(), "bias": bias.float().cpu()}))
                    rec["max_abs_vs_oracle_image0"] = float((out[:1].float().cpu() - want).abs().max())
                print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
