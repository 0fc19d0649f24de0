#!/usr/bin/env python3
"""k_lvit_window (one workgroup per window, q/k/v/attention on chip) against the shipped three-kernel LViT level-1 chain
(k_embed_qkv2 -> k_attention_hm -> k_mlp2) at the encoder shape (B = 8, 24 ch @ 256x256: 512 windows) and the grouped decoder count
(1536 windows, run as 3 x B); cold caches, interleaved rounds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cfen_vit_dehazing_amd import ops, packing
from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.manifest import generate_state_dict

d = "cuda:0"
cfg = NetConfig(24, 4, patch_size=32, load_size=256)
g = cfg.vit("localvit_encoder_01")
sd = {k: (v.half() if v.dtype.is_floating_point else v) for k, v in generate_state_dict(cfg, seed=0, with_dead=False).items() if k.startswith(g.name + ".")}
pk = packing.pack_vit(sd, g, torch.float16)
pk.update(packing.pack_lvit_window(sd, g, torch.float16))
pk = {k: v.to(d).contiguous() for k, v in pk.items()}
n = g.name
flush = torch.empty(320 * 1024 * 1024, dtype=torch.uint8, device=d)


def timed(f):
    flush.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); f(); e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3


for B in (8, 24):
    x = (torch.rand(B, 256, 256, 24, generator=torch.Generator().manual_seed(1)) * 2 - 1).half().to(d)
    nwin = B * 64

    def chain():
        x1, qkv = ops.embed_qkv(x, 24, 32, 2, pk[n + ".embed.wk"], pk[n + ".embed.b"], pk[n + ".pos"], pk[n + ".ln1.g"], pk[n + ".ln1.b"],
                                pk[n + ".qkv.wk"], head_major_heads=4)
        att = ops.attention_head_major(qkv, nwin, 256, 4)
        return ops.mlp_block(x1, pk[n + ".ffn1.wk"], pk[n + ".ffn1.b"], pk[n + ".ffn2.wk"], pk[n + ".ffn2.b"], ln=(pk[n + ".ln2.g"], pk[n + ".ln2.b"]),
                             second=(pk[n + ".head1.wk"], pk[n + ".head1.b"], pk[n + ".head2.wk"], pk[n + ".head2.b"]),
                             proj=(att, pk[n + ".proj.w"]), fold=(B, 256, 256, 24, 24, 32, 2))

    fused = lambda: ops.lvit_window(x, 24, 32, 2, pk, n, g.hidden)
    a, b = chain(), fused()
    torch.cuda.synchronize()
    print("B=%d: fused vs chain max-abs %.3e" % (B, float((a.float() - b.float()).abs().max())))
    tc, tf, tg, th, t5, t0, t8, t9, t6 = [], [], [], [], [], [], [], [], []
    ops.tune("lvit.shape", 4)
    p4 = fused(); torch.cuda.synchronize()
    ops.tune("lvit.shape", 2)
    print("B=%d: hand-pipelined attention reads (lvit.shape 4) bitwise equal to shape 2: %s" % (B, bool(torch.equal(p4, fused()))))
    for _ in range(7):
        tc.append(timed(chain))
        ops.tune("lvit.shape", 2); tf.append(timed(fused))
        ops.tune("lvit.shape", 3); tg.append(timed(fused))
        ops.tune("lvit.shape", 4); th.append(timed(fused))
        ops.tune("lvit.shape", 5); t5.append(timed(fused))
        ops.tune("lvit.shape", 0); t0.append(timed(fused))
        ops.tune("lvit.shape", 8); t8.append(timed(fused))
        ops.tune("lvit.shape", 9); t9.append(timed(fused))
        ops.tune("lvit.shape", 6); t6.append(timed(fused))
    ops.tune("lvit.shape", 1)
    c = fused(); torch.cuda.synchronize()
    ops.tune("lvit.shape", 0)
    print("B=%d: 4-wave variant vs chain max-abs %.3e" % (B, float((a.float() - c.float()).abs().max())))
    tc.sort(); tf.sort(); tg.sort(); th.sort(); t5.sort(); t0.sort(); t8.sort(); t9.sort(); t6.sort()
    ops.tune("lvit.shape", 6); p6 = fused(); torch.cuda.synchronize(); ops.tune("lvit.shape", 2)
    print("B=%d (%d windows): 64-row front chunks (lvit.shape 6) %.1f us against %.1f us, bitwise equal: %s" % (B, nwin, t6[3], tf[3], bool(torch.equal(p6, fused()))))
    print("B=%d (%d windows): timing experiments, results invalid: without the softmax's exp / sum (lvit.shape 8) %.1f us, without the whole softmax (lvit.shape 9) %.1f us" % (B, nwin, t8[3], t9[3]))
    print("B=%d (%d windows): 8 waves x 2 token tiles: compiler-scheduled reads (lvit.shape 0) %.1f us, hand-issued (lvit.shape 5) %.1f us" % (B, nwin, t0[3], t5[3]))
    print("B=%d (%d windows): lvit.shape 2 (compiler-scheduled LDS reads) %.1f us, lvit.shape 4 (hand-issued, 8 K fragments / one V block ahead) %.1f us = %.0f TF/s"
          % (B, nwin, tf[3], th[3], B * 7.95e9 / th[3] / 1e6))
    fl = B * 7.95e9          # 7.95 GFLOP per level-1 instance and image
    print("B=%d (%d windows): chain %.1f us (incl. torch allocations between its 3 launches), window kernel 16 waves, denominator on the vector pipe %.1f us = %.0f TF/s, "
          "16 waves, denominator by MFMA %.1f us = %.0f TF/s" % (B, nwin, tc[3], tf[3], fl / tf[3] / 1e6, tg[3], fl / tg[3] / 1e6))
