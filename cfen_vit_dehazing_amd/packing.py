"""One-time repack of a reference-keyed state_dict into the layouts the HIP kernels read.

Input: the 958-key state_dict of `dec_ipt` (reference models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:104-388,
saved by models/base_model.py:89-101).  Output: {packed name: tensor} with the names/sizes
csrc/cfen_net.cpp expects.  The never-called `decoder.*`, `query_embed`, `sub_mean`, `add_mean`
tensors (208.6 M parameters) are dropped here.

Layout choices:
  * token features are ordered (i, j, c) instead of F.unfold's (c, i, j) (v3:1140): every weight axis
    that lives in the D-sized residual stream is permuted with `token_perm`, so patchify/unpatchify
    are contiguous 16-byte copies of NHWC pixels;
  * conv weights are [phase][Cout_pad][K], K = tap*Cin + c zero-padded to the MFMA chunk, channels
    padded to the map's channel stride; ConvTranspose2d(4,2,1) is split into its 4 output-parity phases;
  * conv bias and ActNorm2d (models/actnorm.py:39-42, y = (x + b) * exp(w)) fold into one fp32
    per-channel (scale, shift) pair.
"""
import torch

from .config import NetConfig, BRANCHES


def round_up(v, m):
    return (v + m - 1) // m * m


def cs_of(c):
    return round_up(c, 8)


def token_perm(C, p):
    """perm[f_new] = f_ref with f_new = (i*p + j)*C + c and f_ref = c*p*p + i*p + j."""
    return torch.arange(C * p * p).view(C, p * p).t().reshape(-1)


def kperm32(n):
    """k-axis permutation that lets an MFMA accumulator tile pair feed the next MFMA as its B operand
    (csrc/k_mlp.hip): inside every group of 32, slot 8h+j holds index 4h+j (j<4) or 16+4h+(j-4) (j>=4)."""
    assert n % 32 == 0
    slot = torch.arange(32)
    hh, jj = slot // 8, slot % 8
    old = torch.where(jj < 4, 4 * hh + jj, 16 + 4 * hh + (jj - 4))
    return (torch.arange(0, n, 32).view(-1, 1) + old.view(1, -1)).reshape(-1)


FUSED_MLP_DIMS = (96, 192)


def mlp_is_fused(g, dtype):
    # LViT only: GViT has too few tokens per image to fill the chip with 128-token workgroups
    return g.kind == "lvit" and g.shrink == 1 and g.dim in FUSED_MLP_DIMS and g.hidden % (64 if dtype == torch.float16 else 32) == 0


def front_is_fused(g):
    """mirror of cfen_embed_qkv_supported (csrc/k_embed.hip): LViT levels 1 and 2 run gather + embedding + LN1 + qkv as one kernel"""
    return g.kind == "lvit" and g.shrink == 1 and g.dim in (96, 192)


def lvit_q_rows(dh=24):
    """Row layout of one head's W_q for k_lvit_window (csrc/k_lvit.hip): 32 rows = two 16-row MFMA tiles (a, b) such that the
    accumulator tile PAIR packs into the next MFMA's B operand with k slot s holding head dim d = s: tile a row 4h+j is d = 8h+j,
    tile b row 4h+j is d = 8h+4+j (h = 0..3, j = 0..3); -1 marks the zero rows of the padding dims d >= dh."""
    rows = []
    for half in (0, 4):
        for r in range(16):
            d = 8 * (r // 4) + half + r % 4
            rows.append(d if d < dh else -1)
    return rows


def window_fusable(g, dtype):
    """mirror of cfen_lvit_window_supported (csrc/k_lvit.hip): LViT level 1 (D = 96, 4 heads of 24, 256-token windows), fp16"""
    # (cfen_net.cpp builds `fused_window` only on top of the fused MLP and the fused front half: fp16 needs hidden % 64 == 0 there)
    return (g.kind == "lvit" and g.shrink == 1 and dtype == torch.float16 and g.dim == 96 and g.heads == 4 and g.seq == 256 and g.hidden % 32 == 0
            and mlp_is_fused(g, dtype) and front_is_fused(g))


def pack_lvit_window(sd, g, dtype):
    """the weight stream of the one-workgroup-per-window LViT kernel: linear_encoding, K / V rows, per-head W_q tiles + out_proj slices, both MLP pairs"""
    n, D, dh = g.name, g.dim, g.dim // g.heads
    perm = token_perm(g.channels, g.patch)
    kp = kperm32(D)
    e = n + ".encoder.layers.0"
    w_in = sd[e + ".self_attn.in_proj_weight"][:, perm][:, kp]            # [3D][D], input features in token order, k axis slotted
    wq_all, wk, wv = w_in[:D], w_in[D:2 * D], w_in[2 * D:]
    rows = lvit_q_rows(dh)
    wq = torch.zeros(g.heads, 32, D, dtype=w_in.dtype, device=w_in.device)
    for hd in range(g.heads):
        for r, d in enumerate(rows):
            if d >= 0:
                wq[hd, r] = wq_all[hd * dh + d]
    w_out = sd[e + ".self_attn.out_proj.weight"][perm]                    # [D][D]: rows in token order, columns = attention features
    slot_d = kperm32(32)
    wp = torch.zeros(g.heads, D, 32, dtype=w_in.dtype, device=w_in.device)
    for hd in range(g.heads):
        for s_, d in enumerate(slot_d.tolist()):
            if d < dh:
                wp[hd, :, s_] = w_out[:, hd * dh + d]
    # ONE stream of 1 KiB MFMA A fragments in the order k_lvit_window consumes them (csrc/k_lvit.hip): lane l of a fragment holds row
    # (l & 15) of its 16-row tile, k elements (l >> 4) * 8 .. + 7 of its 32-wide k-chunk.  A chunk = a 32-row block [32][D] as fragments
    # (u, k) = (row tile, k-chunk), u major ("R1"), for the attention and MLP chunks followed by a [D][32] block as fragments i = row tile ("R2")
    def r1(blk):                                                          # [32][D] -> [2 u][D / 32 k][4 hq][16 r][8 e]
        return blk.reshape(2, 16, D // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)

    def r2(blk):                                                          # [D][32] -> [D / 16 i][4 hq][16 r][8 e]
        return blk.reshape(D // 16, 16, 4, 8).permute(0, 2, 1, 3).reshape(-1)

    kd, kh = kperm32(D), kperm32(g.hidden)
    we = sd[n + ".linear_encoding.weight"][perm][:, perm][:, kd]
    wkv = torch.cat((wk, wv), 0)
    parts = [r1(we[32 * c:32 * c + 32]) for c in range(D // 32)] + [r1(wkv[32 * c:32 * c + 32]) for c in range(2 * D // 32)]
    for hd in range(g.heads):
        parts += [r1(wq[hd]), r2(wp[hd])]
    for a1, a2 in ((e + ".linear1.weight", e + ".linear2.weight"), (n + ".mlp_head.0.weight", n + ".mlp_head.3.weight")):
        w1 = sd[a1][:, perm][:, kd]                                       # [H][D]
        w2 = sd[a2][perm][:, kh]                                          # [D][H], hidden axis slotted like the accumulator pairs it multiplies
        for hc in range(g.hidden // 32):
            parts += [r1(w1[32 * hc:32 * hc + 32]), r2(w2[:, 32 * hc:32 * hc + 32])]
    return {n + ".lw.ws": torch.cat(parts).to(dtype).contiguous()}


def pack_wtile(w):
    """[N][K] row-major -> tile-major [ceil(N / 96)][K * esz / 128][96][128 / esz] flattened (rows past N zero): one K-step of a 96-feature
    tile of k_gemm_dma is one contiguous 12 KB run (csrc/cfen_internal.hpp CfenGemmPtrs::wtile)"""
    n, k = w.shape
    bk = 128 // w.element_size()
    assert k % bk == 0, "tile-major weights need K * sizeof(T) % 128 == 0"
    n96 = round_up(n, 96)
    wp = torch.zeros(n96, k, dtype=w.dtype, device=w.device)
    wp[:n] = w
    return wp.view(n96 // 96, 96, k // bk, bk).permute(0, 2, 1, 3).contiguous().view(-1)


def pack_stream_tiles(w):
    """[N][K] (N % 16 == 0, K % 32 == 0) -> fragment stream [N / 16 feature tiles][K / 32 k-chunks][64 lanes][8]: fragment (i, c) is the 16 x 32 MFMA A
    operand of feature tile i, k-chunk c in lane order (lane l: row i*16 + (l & 15), k = c*32 + (l >> 4)*8 .. +7) -- all k-chunks of a feature tile
    are ONE contiguous run, which is what a wave of csrc/k_gvit.hip streams."""
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0
    return w.reshape(n // 16, 16, k // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def gvit_chain_ok(g, dtype):
    """mirror of cfen_net.cpp Vit::chain: GViT blocks whose GEMMs run as persistent chains (csrc/k_gvit.hip)"""
    return g.kind == "gvit" and g.shrink == 1 and dtype == torch.float16 and g.dim % 128 == 0 and g.hidden % 128 == 0


def pack_stream_sq(w):
    """[N][K] (N % 16 == 0, K % 32 == 0, k axis as the kernel wants it) -> fragment stream [K / 32 phases][N / 16 fragments][64 lanes][8]:
    fragment (c, i) is the 16 x 32 MFMA A operand of feature tile i, k-chunk c in lane order -- lane l holds row i*16 + (l & 15), k = c*32 +
    (l >> 4)*8 .. +7 (csrc/cfen_common.hpp fragment contract).  One phase of csrc/k_stream.hip = all feature tiles of one k-chunk."""
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0
    return w.reshape(n // 16, 16, k // 32, 4, 8).permute(2, 0, 3, 1, 4).contiguous().view(-1)


def pack_stream_pair(w1k, w2k):
    """Linear pair y = W2 relu(W1 x + b1) as ONE fragment stream for csrc/k_stream.hip (k_mlp3): per 32-unit slice t of the hidden dimension
    a W1 phase -- fragments (c, u), u fastest: rows t*32 + u*16 + (l & 15) of w1k [H][D], k-chunk c -- then a W2 phase -- fragments i: rows i*16 + (l & 15)
    of w2k [D][H], k = the slice's 32 hidden units.  Both phases hold D / 16 fragments of 1 KiB.  w1k / w2k carry kperm32 on their k axis."""
    h, d = w1k.shape
    assert w2k.shape == (d, h) and h % 32 == 0 and d % 32 == 0
    nd = d // 16
    # W1 fragments alternate the slice's two 16-row tiles (c-major, u fastest): four independent accumulation chains (2 tiles x 2 token tiles)
    # rotate through the MFMA pipe instead of two -- a dependent 16x16x32 MFMA every second issue slot stalls the pipe (35 % issue-stall
    # cycles in profiles/r03b_sq_wave_states.txt)
    p1 = w1k.reshape(h // 32, 2, 16, d // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).reshape(h // 32, 1, nd, 512)     # [t][c][u][hq][r][e]
    p2 = w2k.reshape(nd, 16, h // 32, 4, 8).permute(2, 0, 3, 1, 4).reshape(h // 32, 1, nd, 512)               # [t][i][hq][r][e]
    return torch.cat((p1, p2), 1).contiguous().view(-1)


def pack_stream_rows(wk):
    """[N][K] (k axis kperm32'd) -> ROW-TILE fragment stream for csrc/k_stream.hip (k_front3): phase t = output rows t*32 .. +31, fragments
    (c, u) = k-chunk c, 16-row tile u (u fastest) in lane order; K / 16 fragments of 1 KiB per phase (the W1 phases of pack_stream_pair on their own)."""
    n, k = wk.shape
    assert n % 32 == 0 and k % 32 == 0
    return wk.reshape(n // 32, 2, 16, k // 32, 4, 8).permute(0, 3, 1, 4, 2, 5).contiguous().view(-1)     # [t][c][u][hq][r][e]: u fastest, as pack_stream_pair


STREAM_MLP_DIMS = (384, 192)


def mlp_is_streamed(g, dtype):
    """mirror of cfen_net.cpp Vit::stream_mlp: LViT levels 3 (D = 384) and 2 (D = 192) run proj + LN2 + FFN + mlp_head + fold as one k_mlp3 launch"""
    return (g.kind == "lvit" and g.shrink == 1 and dtype == torch.float16 and g.dim in STREAM_MLP_DIMS and g.hidden % 32 == 0
            and g.hidden <= 4 * g.dim)


def gvit_is_streamed(g, dtype):
    """mirror of cfen_net.cpp Vit::gstream: GViT blocks of embedding dim 384 (level 1) run on the LViT-3 stream kernels (k_front3 / k_mlp3)"""
    return (g.kind == "gvit" and g.shrink == 1 and dtype == torch.float16 and g.dim == 384 and g.hidden % 32 == 0 and g.hidden <= 4 * g.dim
            and g.channels % 8 == 0)


GVIT_WEIGHT_SUFFIXES = (".embed.w", ".qkv.w", ".qkv.wl", ".proj.w", ".ffn1.w", ".ffn1.wl", ".ffn2.w", ".head1.w", ".head2.w")


def ln_folded(w_packed, gamma, beta, bias, name, dtype, w_full):
    """entries `name`.wl / .s / .bl of a Linear that follows a LayerNorm (gamma, beta): see pack_vit"""
    w64 = w_full.double()
    wl = (w64 * gamma.double()[None, :]).to(dtype)
    bl = w64 @ beta.double()
    if bias is not None:
        bl = bl + bias.double()
    return {name + ".wl": wl.contiguous(), name + ".s": wl.double().sum(1).float(), name + ".bl": bl.float()}


def pack_vit_shrunk(sd, g, dtype):
    """v5 LViT (networks_iid_hlgvit_crs_gd4_cfs_v5.py:1086-1127): the block runs on C/4 = 6 / 12 / 24 channels.  Those maps live at a
    channel stride of 8 / 16 / 24 (16-byte pixel vectors), so a token row has D_pad = p*p*cs slots of which D = p*p*C are real, and the
    6-wide heads are padded to 8.  Every matrix is scattered into its zero-padded layout here, once:
      residual-stream axis: ref feature f = c*p*p + i*p + j  ->  slot (i*p + j)*cs + c          (padding slots: zero rows / columns,
                                                                                                 gamma = beta = bias = 0)
      attention axis:       ref feature a = head*dh + d      ->  slot head*dh_pad + d
    and 1/sqrt(dh) = sqrt(dh_pad/dh) / sqrt(dh_pad): the kernel scales by 1/sqrt(dh_pad), the factor sqrt(dh_pad/dh) rides on W_q.
    csrc/cfen_net.cpp (Vit::Dn, Da, dh) normalises over the D real entries and runs the attention kernel at head_dim dh_pad."""
    import math
    n, C, p, D, H, heads = g.name, g.channels, g.patch, g.dim, g.hidden, g.heads
    cs = cs_of(C)
    Dp = p * p * cs
    dh = D // heads
    dhp = round_up(dh, 8)
    Da = heads * dhp
    f = torch.arange(D)
    c, ij = f // (p * p), f % (p * p)
    pos = ij * cs + c                                              # slot of ref feature f in the padded token row
    a = torch.arange(D)
    apos = (a // dh) * dhp + a % dh                                # slot of attention feature a
    e = n + ".encoder.layers.0"
    f32 = torch.float32
    dev = sd[n + ".linear_encoding.weight"].device
    pos, apos = pos.to(dev), apos.to(dev)

    def mat(w, rows, nrows, cols, ncols):
        out = torch.zeros(nrows, ncols, dtype=torch.float32, device=dev)
        r = rows if rows is not None else torch.arange(nrows, device=dev)
        cidx = cols if cols is not None else torch.arange(ncols, device=dev)
        out[r[:, None], cidx[None, :]] = w.float()
        return out.to(dtype).contiguous()

    def vec(v, idx, nrows):
        out = torch.zeros(nrows, dtype=f32, device=dev)
        out[idx] = v.float()
        return out

    w_in = sd[e + ".self_attn.in_proj_weight"].float()
    qkv = torch.zeros(3 * Da, Dp, dtype=torch.float32, device=dev)
    for t in range(3):
        blk = w_in[t * D:(t + 1) * D] * (math.sqrt(dhp / dh) if t == 0 else 1.0)
        qkv[(t * Da + apos)[:, None], pos[None, :]] = blk
    return {
        n + ".embed.w": mat(sd[n + ".linear_encoding.weight"], pos, Dp, pos, Dp), n + ".embed.b": vec(sd[n + ".linear_encoding.bias"], pos, Dp),
        n + ".pos": mat(sd[n + ".position_encoding.pe.weight"][:g.seq], None, g.seq, pos, Dp),
        n + ".ln1.g": vec(sd[e + ".norm1.weight"], pos, Dp), n + ".ln1.b": vec(sd[e + ".norm1.bias"], pos, Dp),
        n + ".qkv.w": qkv.to(dtype).contiguous(),
        n + ".proj.w": mat(sd[e + ".self_attn.out_proj.weight"], pos, Dp, apos, Da),
        n + ".ln2.g": vec(sd[e + ".norm2.weight"], pos, Dp), n + ".ln2.b": vec(sd[e + ".norm2.bias"], pos, Dp),
        n + ".ffn1.w": mat(sd[e + ".linear1.weight"], None, H, pos, Dp), n + ".ffn1.b": sd[e + ".linear1.bias"].to(f32),
        n + ".ffn2.w": mat(sd[e + ".linear2.weight"], pos, Dp, None, H), n + ".ffn2.b": vec(sd[e + ".linear2.bias"], pos, Dp),
        n + ".head1.w": mat(sd[n + ".mlp_head.0.weight"], None, H, pos, Dp), n + ".head1.b": sd[n + ".mlp_head.0.bias"].to(f32),
        n + ".head2.w": mat(sd[n + ".mlp_head.3.weight"], pos, Dp, None, H), n + ".head2.b": vec(sd[n + ".mlp_head.3.bias"], pos, Dp),
    }


def pack_vit(sd, g, dtype):
    if g.shrink > 1:
        out = pack_vit_shrunk(sd, g, dtype)
        # where the padded row has no padding (24 channels) the LayerNorm fold of the unfused path applies as everywhere else
        if cs_of(g.channels) == g.channels and (g.dim * (2 if dtype == torch.float16 else 4)) % 128 == 0:
            n, e = g.name, g.name + ".encoder.layers.0"
            out.update(ln_folded(None, out[n + ".ln1.g"], out[n + ".ln1.b"], None, n + ".qkv", dtype, out[n + ".qkv.w"].float()))
            out.update(ln_folded(None, out[n + ".ln2.g"], out[n + ".ln2.b"], out[n + ".ffn1.b"], n + ".ffn1", dtype, out[n + ".ffn1.w"].float()))
        return out
    n = g.name
    perm = token_perm(g.channels, g.patch)
    e = n + ".encoder.layers.0"
    f32 = torch.float32
    out = {
        n + ".embed.w": sd[n + ".linear_encoding.weight"][perm][:, perm].to(dtype),
        n + ".embed.b": sd[n + ".linear_encoding.bias"][perm].to(f32),
        n + ".pos": sd[n + ".position_encoding.pe.weight"][:g.seq][:, perm].to(dtype),
        n + ".ln1.g": sd[e + ".norm1.weight"][perm].to(f32), n + ".ln1.b": sd[e + ".norm1.bias"][perm].to(f32),
        n + ".qkv.w": sd[e + ".self_attn.in_proj_weight"][:, perm].to(dtype),
        n + ".proj.w": sd[e + ".self_attn.out_proj.weight"][perm].to(dtype),
        n + ".ln2.g": sd[e + ".norm2.weight"][perm].to(f32), n + ".ln2.b": sd[e + ".norm2.bias"][perm].to(f32),
        n + ".ffn1.w": sd[e + ".linear1.weight"][:, perm].to(dtype), n + ".ffn1.b": sd[e + ".linear1.bias"].to(f32),
        n + ".ffn2.w": sd[e + ".linear2.weight"][perm].to(dtype), n + ".ffn2.b": sd[e + ".linear2.bias"][perm].to(f32),
        n + ".head1.w": sd[n + ".mlp_head.0.weight"][:, perm].to(dtype), n + ".head1.b": sd[n + ".mlp_head.0.bias"].to(f32),
        n + ".head2.w": sd[n + ".mlp_head.3.weight"][perm].to(dtype), n + ".head2.b": sd[n + ".mlp_head.3.bias"][perm].to(f32),
    }
    # LayerNorm folded into the following GEMM (csrc/k_gemm.hip: CfenGemmPtrs::lnf_s) wherever the block is not a fused kernel:
    # LN(x) W^T + b = rstd (x (W gamma)^T - mean s) + (W beta + b) with s = row sums of the ROUNDED W gamma (so the mean cancels exactly)
    # (k_gemm_dma stages whole 128-byte K steps: cfen_net.cpp Vit::ln_fold1 / ln_fold2 ask for the folded entries only then)
    k128 = (g.dim * (2 if dtype == torch.float16 else 4)) % 128 == 0
    if not front_is_fused(g) and k128:
        out.update(ln_folded(out[n + ".qkv.w"], out[n + ".ln1.g"], out[n + ".ln1.b"], None, n + ".qkv", dtype,
                             sd[e + ".self_attn.in_proj_weight"][:, perm]))
    if not mlp_is_fused(g, dtype) and k128:
        out.update(ln_folded(out[n + ".ffn1.w"], out[n + ".ln2.g"], out[n + ".ln2.b"], out[n + ".ffn1.b"], n + ".ffn1", dtype,
                             sd[e + ".linear1.weight"][:, perm]))
    if front_is_fused(g):
        # same matrices, k axis re-slotted so that accumulator tile pairs feed the MFMA directly (fp16 only)
        # (the plain layout stays: "net.fused_front" can switch the fused kernel off per embedding dim)
        for nm in ("embed", "qkv"):
            w = out[n + "." + nm + ".w"]
            out[n + "." + nm + ".wk"] = (w[:, kperm32(g.dim)] if dtype == torch.float16 else w).contiguous()
    if window_fusable(g, dtype):
        out.update(pack_lvit_window(sd, g, dtype))
    if gvit_is_streamed(g, dtype):
        # the same fragment streams for a GViT block of dim 384 ("net.gvit_stream": pooled map -> k_front3 -> attention -> k_mlp3); the row-major /
        # tile-major matrices stay for the launch-per-GEMM plan
        kd, kh = kperm32(g.dim), kperm32(g.hidden)
        out[n + ".proj.ws"] = pack_stream_sq(out[n + ".proj.w"])
        out[n + ".ffn.ws"] = pack_stream_pair(out[n + ".ffn1.w"][:, kd], out[n + ".ffn2.w"][:, kh])
        out[n + ".head.ws"] = pack_stream_pair(out[n + ".head1.w"][:, kd], out[n + ".head2.w"][:, kh])
        out[n + ".embed.ws"] = pack_stream_rows(out[n + ".embed.w"][:, kd])
        out[n + ".qkv.ws"] = pack_stream_rows(out[n + ".qkv.w"][:, kd])
    if mlp_is_streamed(g, dtype):
        # k_mlp3 (csrc/k_stream.hip): out_proj, FFN pair and mlp_head pair as fragment streams; the row-major matrices stay for "net.stream_mlp" = 0
        kd, kh = kperm32(g.dim), kperm32(g.hidden)
        out[n + ".proj.ws"] = pack_stream_sq(out[n + ".proj.w"])
        out[n + ".ffn.ws"] = pack_stream_pair(out[n + ".ffn1.w"][:, kd], out[n + ".ffn2.w"][:, kh])
        out[n + ".head.ws"] = pack_stream_pair(out[n + ".head1.w"][:, kd], out[n + ".head2.w"][:, kh])
        if g.dim == 384:
            # k_front3: linear_encoding and in_proj as row-tile streams (the LayerNorm-folded and plain matrices stay for "net.stream_front" = 0)
            out[n + ".embed.ws"] = pack_stream_rows(out[n + ".embed.w"][:, kd])
            out[n + ".qkv.ws"] = pack_stream_rows(out[n + ".qkv.w"][:, kd])
    if mlp_is_fused(g, dtype):
        # the fused MLP kernel replaces the four separate GEMMs: same matrices, k axis re-slotted (fp16 only)
        for a, b in (("ffn1", "ffn2"), ("head1", "head2")):
            w1, w2 = out.pop(n + "." + a + ".w"), out.pop(n + "." + b + ".w")
            if dtype == torch.float16:
                w1, w2 = w1[:, kperm32(g.dim)], w2[:, kperm32(g.hidden)]
            out[n + "." + a + ".wk"], out[n + "." + b + ".wk"] = w1.contiguous(), w2.contiguous()
    return out


def pack_conv_weight(w, cin_pad, kc, dtype):
    """Conv2d weight (Cout, Cin_total, k, k) with Cin_total = nsrc*Cin -> [1][Cout_pad][Kpad]; taps are
    (src, ky, kx) so a 1x1 conv over a concat keeps its natural channel order."""
    cout, cin_total, k, _ = w.shape
    if k == 1 and cin_total < cin_pad:
        flat = torch.zeros(cout, cin_pad, dtype=w.dtype, device=w.device)   # one source at a padded channel stride (v5 conv_extend: 6 -> 8)
        flat[:, :cin_total] = w.reshape(cout, cin_total)
    elif k == 1:
        nsrc_cin = cin_total                                   # concat sources are unpadded (C % 8 == 0)
        flat = w.reshape(cout, nsrc_cin)
    else:
        wp = torch.zeros(cout, k, k, cin_pad, dtype=w.dtype, device=w.device)
        wp[..., :cin_total] = w.permute(0, 2, 3, 1)
        flat = wp.reshape(cout, k * k * cin_pad)
    cout_pad, kpad = round_up(cout, 16), round_up(flat.shape[1], kc)
    out = torch.zeros(1, cout_pad, kpad, dtype=dtype, device=w.device)
    out[0, :cout, :flat.shape[1]] = flat.to(dtype)
    return out


def conv_uses_rows_layout(dtype, k, stride, pad, nsrc, cs_in, cout, H, W):
    """mirror of cfen_conv_tile_supported (csrc/k_conv_tile.hip): which Conv2d layers run on the LDS-tiled kernel"""
    pixb = cs_in * (2 if dtype == torch.float16 else 4)
    if stride != 1 or nsrc != 1 or pad != k // 2 or round_up(cout, 16) != 16 or H % 8 or W % 64:
        return False
    if dtype == torch.float16:
        return (pixb == 16 and k == 5) or (pixb == 32 and k in (3, 7))
    return (pixb == 32 and k == 5) or (pixb == 64 and k in (3, 7))


def conv_uses_toeplitz7(dtype, k, stride, pad, nsrc, cs_in, cout, nchw_f32, H, W):
    """mirror of cfen_conv7_tz_supported (csrc/k_conv_tile.hip: k_conv7_tz)"""
    return (dtype == torch.float16 and k == 7 and stride == 1 and pad == 3 and nsrc == 1 and cs_in == 16 and 1 <= cout <= 4 and nchw_f32
            and H % 16 == 0 and W % 64 == 0)


def pack_conv7_toeplitz(w, dtype):
    """Conv2d weight (Cout<=4, Cin<=16, 7, 7) -> [1][16][7*10*16]: row co*4 + dxo holds w[co][:, ky, kx' - dxo] at tap kx' (zero outside),
    so one MFMA column stands for 4 adjacent output pixels."""
    cout, cin = w.shape[0], w.shape[1]
    wz = torch.zeros(4, 4, 7, 10, 16, dtype=dtype, device=w.device)
    wp = w.permute(0, 2, 3, 1).to(dtype)          # (co, ky, kx, ci)
    for dxo in range(4):
        wz[:cout, dxo, :, dxo:dxo + 7, :cin] = wp
    return wz.reshape(1, 16, 7 * 10 * 16)


def head5_supported(dtype, cout, H, W):
    """mirror of cfen_head5_supported (csrc/k_head5.hip)"""
    return dtype == torch.float16 and round_up(cout, 16) == 16 and cs_of(cout) == 16 and H % 8 == 0 and W % 64 == 0


def pack_head5(w, dtype):
    """Conv2d weight (Cout <= 16, 3, 5, 5) -> [16][5 dy][8 taps][4 c]: the layout of csrc/k_head5.hip (8-byte pixels: 3 channels + a zero; a kernel row
    is one 64-byte MFMA chunk of 8 taps, taps 5..7 zero)"""
    cout, cin, k, _ = w.shape
    assert cin == 3 and k == 5 and cout <= 16
    wp = torch.zeros(16, 5, 8, 4, dtype=dtype, device=w.device)
    wp[:cout, :, :5, :3] = w.permute(0, 2, 3, 1).to(dtype)
    return wp.reshape(-1)


def pack_conv_weight_rows(w, cin_pad, dtype):
    """Conv2d weight (Cout<=16, Cin, k, k) -> [1][16][k*KSP*cin_pad]: tap-major, every kernel row padded with zero
    taps to a whole number of 64-byte chunks (KSP taps)."""
    cout, cin, k, _ = w.shape
    pixb = cin_pad * (2 if dtype == torch.float16 else 4)
    tpc = 64 // pixb
    ksp = (k + tpc - 1) // tpc * tpc
    wp = torch.zeros(16, k, ksp, cin_pad, dtype=dtype, device=w.device)
    wp[:cout, :, :k, :cin] = w.permute(0, 2, 3, 1).to(dtype)
    return wp.reshape(1, 16, k * ksp * cin_pad)


_KY = ((1, 3), (0, 2))    # [output parity][tap] -> kernel index; input offsets (0,-1),(+1,0): csrc/cfen_conv.hpp


def pack_convT_weight(w, cin_pad, kc, dtype):
    """ConvTranspose2d weight (Cin, Cout, 4, 4) -> [4 phases][Cout_pad][Kpad], K = (ty*2+tx)*Cin + ci."""
    cin, cout = w.shape[0], w.shape[1]
    cout_pad, kpad = round_up(cout, 16), round_up(4 * cin_pad, kc)
    out = torch.zeros(4, cout_pad, kpad, dtype=dtype, device=w.device)
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    t = ty * 2 + tx
                    out[py * 2 + px, :cout, t * cin_pad:t * cin_pad + cin] = w[:, :, _KY[py][ty], _KY[px][tx]].t().to(dtype)
    return out


def convT_uses_rows_layout(dtype, cin_pad, cout, Hin, Win):
    """mirror of cfen_convT_tile_supported (csrc/k_conv_tile.hip)"""
    if dtype != torch.float16 or Hin % 4 or Win % 32:
        return False
    pixb = (cin_pad * 2 + 63) // 64 * 64
    return (pixb, round_up(cout, 16)) in ((64, 16), (128, 32), (192, 48))


def pack_convT_weight_rows(w, cin_pad, dtype):
    """ConvTranspose2d weight (Cin, Cout, 4, 4) -> [4 phases][Cout_pad][4 taps * CP]: as pack_convT_weight with every
    tap's channels zero-padded to a whole number of 64-byte chunks (CP elements)."""
    cin, cout = w.shape[0], w.shape[1]
    esz = 2 if dtype == torch.float16 else 4
    cp = (cin_pad * esz + 63) // 64 * 64 // esz
    out = torch.zeros(4, round_up(cout, 16), 4, cp, dtype=dtype, device=w.device)
    for py in range(2):
        for px in range(2):
            for ty in range(2):
                for tx in range(2):
                    out[py * 2 + px, :cout, ty * 2 + tx, :cin] = w[:, :, _KY[py][ty], _KY[px][tx]].t().to(dtype)
    return out.reshape(4, out.shape[1], 4 * cp)


def affine(bias, an_w=None, an_b=None, cout_pad=None):
    """(scale, shift) so that y = acc*scale + shift == ActNorm(conv + bias)."""
    bias = bias.float()
    if an_w is None:
        scale, shift = torch.ones_like(bias), bias
    else:
        scale = torch.exp(an_w.float())
        shift = (bias + an_b.float()) * scale
    s = torch.zeros(cout_pad, dtype=torch.float32, device=bias.device)
    t = torch.zeros(cout_pad, dtype=torch.float32, device=bias.device)
    s[:bias.numel()], t[:bias.numel()] = scale, shift
    return s, t


def _actnorm_ready(sd, prefix):
    return int(sd[prefix + ".initialized"]) == 1


def pack_state_dict(sd, cfg: NetConfig, dtype=torch.float16, pending=None, wtile=False, gvit_stream=False):
    """reference state_dict (tensors on any one device) -> packed tensors on the same device.

    An ActNorm2d whose `initialized` buffer is 0 holds no parameters yet: the reference fills it from the statistics of its first
    batch (models/actnorm.py:25-37).  Such a layer gets a neutral epilogue table here and is reported in `pending`
    ({packed layer name: (ActNorm key prefix, conv bias padded to Cout_pad)}); hipnet.dec_ipt hands those layers to
    cfen_net_actnorm_pending, and the first forward initialises them on the device.  Without a `pending` dict this raises."""
    kc = 32 if dtype == torch.float16 else 16
    nf, h = cfg.n_feats, cfg.head_channels
    cfs = cfg.full_res              # sibling generators networks_iid_hlgvit_crs_gd4_cfs.py / ..._crs_gd4.py: see config.NetConfig.image_size
    crs = cfg.variant == "crs"
    out = {}
    for g in cfg.vit_instances():
        pv = pack_vit(sd, g, dtype)
        if gvit_stream and gvit_chain_ok(g, dtype):
            # the persistent chains (csrc/k_gvit.hip) read every matrix as a fragment stream; qkv / ffn1 in their LayerNorm-folded form.  The
            # row-major / tile-major copies stay: "net.gvit_chain" = 0 runs the launch-per-GEMM plan on them
            for nm, src in (("embed", ".embed.w"), ("qkv", ".qkv.wl"), ("proj", ".proj.w"), ("ffn1", ".ffn1.wl"), ("ffn2", ".ffn2.w"), ("head1", ".head1.w"),
                            ("head2", ".head2.w")):
                pv[g.name + "." + nm + ".wf"] = pack_stream_tiles(pv[g.name + src].contiguous())
        if wtile and g.kind == "gvit":      # the GViT GEMMs stream their weights from HBM once per forward: tile-major (pack_wtile)
            for k in list(pv):
                if k.endswith(GVIT_WEIGHT_SUFFIXES):
                    pv[k] = pack_wtile(pv[k].contiguous())
        out.update(pv)

    full = cfg.image_size

    def conv(name, key, cin, an=None, rows=None):
        # rows = (stride, pad) of a full-resolution layer that may run on the LDS-tiled kernel
        w = sd[key + ".weight"]
        cp = round_up(w.shape[0], 16)
        if rows and name.endswith(".conv7") and conv_uses_toeplitz7(dtype, w.shape[2], rows[0], rows[1], 1, cs_of(cin), w.shape[0], True, full, full):
            out[name + ".wz"] = pack_conv7_toeplitz(w, dtype)
        elif rows and conv_uses_rows_layout(dtype, w.shape[2], rows[0], rows[1], 1, cs_of(cin), w.shape[0], full, full):
            out[name + ".wr"] = pack_conv_weight_rows(w, cs_of(cin), dtype)
        else:
            out[name + ".w"] = pack_conv_weight(w, cs_of(cin), kc, dtype)
        out[name + ".scale"], out[name + ".shift"] = epilogue(name, key, an, cp)

    def epilogue(name, key, an, cp):
        if an and not _actnorm_ready(sd, an):
            if pending is None:
                raise NotImplementedError("ActNorm2d '%s' is not initialised (models/actnorm.py:25-37 fills it from the first batch); "
                                          "pack with a `pending` dict so that the first forward initialises it on the device" % an)
            s, t = affine(sd[key + ".bias"], cout_pad=cp)
            pending[name] = (an, t.clone())
            return s, t
        if an:
            return affine(sd[key + ".bias"], sd[an + ".weight"], sd[an + ".bias"], cp)
        return affine(sd[key + ".bias"], cout_pad=cp)

    def convT(name, key, cin, an=None, edge=None):
        w = sd[key + ".weight"]
        cp = round_up(w.shape[1], 16)
        if convT_uses_rows_layout(dtype, cs_of(cin), w.shape[1], edge, edge):
            out[name + ".wr"] = pack_convT_weight_rows(w, cs_of(cin), dtype)
        else:
            out[name + ".w"] = pack_convT_weight(w, cs_of(cin), kc, dtype)
        out[name + ".scale"], out[name + ".shift"] = epilogue(name, key, an, cp)

    for g in cfg.vit_instances():
        if g.shrink > 1:            # v5:1101-1104 conv_shrink / conv_extend (Conv2d 1x1 + ActNorm2d + ReLU)
            conv(g.name + ".shrink", g.name + ".conv_shrink.0", g.map_channels, an=g.name + ".conv_shrink.1")
            conv(g.name + ".extend", g.name + ".conv_extend.0", g.channels, an=g.name + ".conv_extend.1")
    conv("head.0.0", "head.0.0", 3, rows=(1, 2))
    if head5_supported(dtype, sd["head.0.0.weight"].shape[0], full, full):
        out["head.0.0.w5"] = pack_head5(sd["head.0.0.weight"], dtype)
    conv("head.0.1.body.0", "head.0.1.body.0", h, rows=(1, 1))
    conv("head.0.1.body.2", "head.0.1.body.2", h, rows=(1, 1))
    if not cfs:
        conv("ds_conv_e01", "ds_conv_e01.0", h)
    conv("ds_conv_e02", "ds_conv_e02.0", nf)
    conv("ds_conv_e03", "ds_conv_e03.0", 2 * nf)
    for l in (1, 2, 3):
        n = "lgcat_conv_e0%d" % l
        conv(n, n + ".0", nf << (l - 1), an=n + ".1")
    for b in BRANCHES:
        for l in (1, 2, 3):
            n = "lgcat_conv_d0%d%s" % (l, b)
            conv(n, n + ".0", nf << (l - 1), an=n + ".1")
        convT("us_conv_d03" + b, "us_conv_d03%s.0" % b, 4 * nf, edge=cfg.load_size // 4)
        convT("us_conv_d02" + b, "us_conv_d02%s.0" % b, 2 * nf, an="us_conv_d02%s.1" % b, edge=cfg.load_size // 2)
        if not cfs:
            convT("us_conv_d01" + b, "us_conv_d01%s.0" % b, nf, an="us_conv_d01%s.1" % b, edge=cfg.load_size)
        if b != "d" or crs:         # crs:327-330: D's skip fuse is a 1x1 conv too, over three maps
            conv("sk_conv_d03" + b, "sk_conv_d03%s.0" % b, 2 * nf, an="sk_conv_d03%s.1" % b)
            conv("sk_conv_d02" + b, "sk_conv_d02%s.0" % b, nf, an="sk_conv_d02%s.1" % b)
        T = "tail_" + b.upper()
        src = T if not cfs else ("tail_gray" if b == "s" else "tail_color")   # cfs: R and D share tail_color (cfs:334-345, 669, 977)
        if cfs and b == "d":
            # the same module as tail_R: the packed entries are aliases ("@name"), so a device-side ActNorm initialisation of
            # tail_R.conv3's epilogue table (first forward, models/actnorm.py:25-37) is what tail_D.conv3 reads too
            for part in (".conv3.w", ".conv3.wr", ".conv3.scale", ".conv3.shift", ".conv7.w", ".conv7.wr", ".conv7.wz", ".conv7.scale", ".conv7.shift"):
                if "tail_R" + part in out:
                    out[T + part] = "@tail_R" + part
        elif b == "s":
            conv(T + ".conv3", src + ".0.1", h, rows=(1, 1))
            conv(T + ".conv7", src + ".0.4", h, rows=(1, 3))
        else:
            conv(T + ".conv3", src + ".0.1", h, an=src + ".0.2", rows=(1, 1))
            conv(T + ".conv7", src + ".0.5", h, rows=(1, 3))
    for n in (() if crs else ("cfsm2g_d03d", "cfsm2g_d02d")):
        parts = []
        for fc in ("fc_avg_cf1", "fc_avg_cf2", "fc_max_cf1", "fc_max_cf2"):
            parts.append(sd["%s.0.%s.0.weight" % (n, fc)].float().reshape(-1))
            parts.append(sd["%s.0.%s.2.weight" % (n, fc)].float().reshape(-1))
        out[n + ".w"] = torch.cat(parts)
    return {k: (v if isinstance(v, str) else v.contiguous()) for k, v in out.items()}
