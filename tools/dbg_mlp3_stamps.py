#!/usr/bin/env python3
"""k_mlp3<24, ...> (D = 384, H = 1536: LViT level 3 / GViT level 1) -- where a double phase's time goes, and the round-6 issue variants.

For 16 / 64 / 192 workgroups of 128 tokens, cold caches (a 256 MiB fill between calls): launch time of the shipped kernel and of each `mlp3.debug` variant,
results compared bit for bit with the shipped kernel's; then the STAMPED builds (s_memtime around the DMA wait, the barrier and the phase body of workgroup 0,
printed by the library on stderr).  Usage: dbg_mlp3_stamps.py [variants...]  (default: all)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing

d = "cuda:0"
flush = torch.empty(256 << 20, dtype=torch.uint8, device=d)
NAMES = {0: "shipped default", 100: "burst issue (rounds 3-5)", 1: "no DMA refills (invalid)", 2: "no MFMAs in the hidden loop (invalid)", 8: "8 waves x 1 tile", 11: "spread issue",
         18: "refills in thirds, 4 reads ahead", 19: "... + early repack (shipped)"}
STAMPED = {64: "burst", 67: "burst, no refills", 68: "no LDS fragment reads", 69: "no reads, no refills", 70: "4-byte refill pieces", 71: "refills + barrier only"}


def timeit(f, n=12):
    f(); f()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        flush.fill_(1)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); f(); e.record(); torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[0]


def main():
    want = [int(v) for v in sys.argv[1:]] or [100, 11, 18, 1, 2]
    D, H = 384, 1536
    torch.manual_seed(0)
    for M in (2048, 8192, 24576):
        r = lambda *s, sc=1.0: (torch.randn(*s, device=d) * sc).half()
        x, att = r(M, D), r(M, D)
        wp, w1a, w2a, w1b, w2b = r(D, D, sc=D ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5), r(H, D, sc=D ** -0.5), r(D, H, sc=H ** -0.5)
        g, b = torch.ones(D, device=d), torch.zeros(D, device=d)
        b1, b2 = torch.randn(H, device=d) * 0.1, torch.randn(D, device=d) * 0.1
        kd, kh = packing.kperm32(D).to(d), packing.kperm32(H).to(d)
        sa, sb, sp = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]), packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]), packing.pack_stream_sq(wp)
        fl = (8.0 * D * H + 2.0 * D * D) * M
        call = lambda: ops.mlp_stream_block(x, sa, b1, b2, H, ln=(g, b), second=(sb, b1, b2), proj=(att, sp))
        ops.tune("mlp3.debug", 0)
        ref = call().clone()
        med, best = timeit(call)
        print("M=%d (%d workgroups): shipped %.1f us median / %.1f best = %.0f TF" % (M, M // 128, med, best, fl / med / 1e6), flush=True)
        for v in want:
            try:
                ops.tune("mlp3.debug", v)
                out = call()
            except Exception as ex:  # a variant this build does not carry
                print("  debug=%d: %s" % (v, ex))
                continue
            same = bool(torch.equal(out, ref))
            med, best = timeit(call)
            print("  debug=%-3d %-40s %.1f us median / %.1f best = %.0f TF   bitwise equal to shipped: %s" % (v, NAMES.get(v, "?"), med, best, fl / med / 1e6, same), flush=True)
        if M in (2048, 24576):
            for v, nm in STAMPED.items():
                sys.stderr.write("==== M=%d stamped build %d (%s)\n" % (M, v, nm))
                sys.stderr.flush()
                try:
                    ops.tune("mlp3.debug", v)
                    for _ in range(3):
                        flush.fill_(1)
                        call()
                        torch.cuda.synchronize()
                except Exception as ex:
                    sys.stderr.write("  %s\n" % ex)
        ops.tune("mlp3.debug", 0)


if __name__ == "__main__":
    main()
