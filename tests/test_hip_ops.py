"""GPU parity of every HIP operator (called through the C ABI) against the CPU oracle / float64 math.

Tolerances: CFEN_F32 runs exact-fp32 MFMA -> 2e-5 * scale; CFEN_F16 stores fp16 and accumulates fp32 ->
inputs are rounded to fp16 first and the result may differ by fp16 output rounding (2^-10 relative)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cfen_oracle
from cfen_vit_dehazing_amd import ops, packing

pytestmark = pytest.mark.gpu
DTYPES = [torch.float32, torch.float16]


def dev():
    return torch.device("cuda:0")


def tol(dtype, scale=1.0):
    return (3e-5 if dtype == torch.float32 else 3e-3) * scale


def rnd(shape, seed, dtype, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def close(got, want, atol, what=""):
    d = float((got.double().cpu() - want.double().cpu()).abs().max())
    assert d <= atol, "%s max-abs %.3e > %.1e" % (what, d, atol)
    return d


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_exact_integers_asymmetric(dtype):
    # exact in both dtypes: catches any transposed / permuted fragment mapping
    M, N, K = 37, 48, 64
    x = (torch.arange(M * K).view(M, K) % 7 - 3).to(dtype)
    w = ((torch.arange(N * K).view(N, K) * 5) % 11 - 5).to(dtype)
    y = ops.gemm_nt(x.to(dev()), w.to(dev()))
    assert torch.equal(y.float().cpu(), x.float() @ w.float().t())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(16, 96, 96), (100, 288, 96), (300, 128, 192), (128, 384, 1536), (257, 96, 384), (2048, 768, 192)])
def test_gemm_epilogues(dtype, M, N, K):
    x, w = rnd((M, K), 1, dtype), rnd((N, K), 2, dtype, 1 / math.sqrt(K))
    bias = rnd((N,), 3, torch.float32)
    res = rnd((M, N), 4, dtype)
    S = 16
    pos = rnd((S, N), 5, dtype)
    ref = x.double() @ w.double().t()
    d = dev()
    close(ops.gemm_nt(x.to(d), w.to(d)), ref, tol(dtype, 4), "plain")
    want = torch.relu(ref + bias.double()) + res.double() + pos.double()[torch.arange(M) % S]
    got = ops.gemm_nt(x.to(d), w.to(d), bias=bias.to(d), residual=res.to(d), pos=pos.to(d), relu=True)
    close(got, want, tol(dtype, 8), "bias+relu+res+pos")
    # in-place residual (Y aliases R), as the transformer block uses it
    r = res.to(d).clone()
    ops.gemm_nt(x.to(d), w.to(d), residual=r, out=r)
    close(r, ref + res.double(), tol(dtype, 8), "in-place residual")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,D", [(5, 96), (64, 192), (33, 384), (16, 1536), (7, 2048)])
def test_layernorm(dtype, M, D):
    x = rnd((M, D), 1, dtype, 2.0) + 0.5
    g, b = 1 + 0.1 * rnd((D,), 2, torch.float32), 0.1 * rnd((D,), 3, torch.float32)
    want = cfen_oracle.layer_norm(x.double(), g.double(), b.double())
    close(ops.layernorm(x.to(dev()), g.to(dev()), b.to(dev())), want, tol(dtype, 4))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,D,N,relu", [(2048, 384, 1152, 0), (100, 384, 768, 1), (32, 1536, 4608, 0), (7, 3072, 96, 1), (300, 6144, 192, 1)])
def test_gemm_with_layernorm_folded(dtype, M, D, N, relu):
    """norm1 -> in_proj and norm2 -> linear1 (v3:1383-1389) as ONE GEMM on the un-normalised rows (cfen_gemm_ln, packing.ln_folded);
    rows carry a mean of half their spread, like the residual stream, so the mean * s cancellation is exercised"""
    from cfen_vit_dehazing_amd.packing import ln_folded
    x = rnd((M, D), 1, dtype, 2.0) + 1.0
    w = rnd((N, D), 2, torch.float32, D ** -0.5)
    g, b, bias = 1 + 0.1 * rnd((D,), 3, torch.float32), 0.1 * rnd((D,), 4, torch.float32), rnd((N,), 5, torch.float32)
    want = cfen_oracle.layer_norm(x.double(), g.double(), b.double()) @ w.double().t() + bias.double()
    if relu:
        want = want.relu()
    f = ln_folded(None, g, b, bias, "l", dtype, w)
    got = ops.gemm_ln(x.to(dev()), f["l.wl"].to(dev()), f["l.s"].to(dev()), f["l.bl"].to(dev()), relu=bool(relu))
    close(got, want, tol(dtype, 8))
    if D > 2048:   # cfen_layernorm's own limit (the generator's widest token is 1536)
        return
    # the same numbers through the two-launch path the fold replaces
    two = ops.gemm_nt(ops.layernorm(x.to(dev()), g.to(dev()), b.to(dev())), w.to(dtype).to(dev()), bias.to(dev()), relu=bool(relu))
    close(two, want, tol(dtype, 8))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K,nsplit", [(128, 1536, 1536, 4), (128, 1536, 6144, 8), (100, 200, 768, 2), (512, 768, 3072, 8), (33, 96, 256, 2), (16, 4608, 1536, 3)])
def test_gemm_split_k_in_launch_reduction(dtype, M, N, K, nsplit):
    """k_gemm_dma with K cut into slices and the last-arriving workgroup of a tile adding the partial tiles in slice order: against fp64,
    every epilogue (bias / ReLU / residual / folded LayerNorm), bit-reproducible over repeated launches on one scratch buffer, and the arrival
    counters are back to zero after every call (ragged token / feature tiles included)"""
    d = dev()
    x, w = rnd((M, K), 1, dtype), rnd((N, K), 2, dtype, 1 / math.sqrt(K))
    bias, res = rnd((N,), 3, torch.float32), rnd((M, N), 4, dtype)
    tiles = ((N + 95) // 96) * ((M + 31) // 32)
    scratch = torch.zeros(4096 + tiles * nsplit * 14336, dtype=torch.uint8, device=d)
    ref = x.double() @ w.double().t()
    got = ops.gemm_splitk(x.to(d), w.to(d), nsplit, scratch=scratch)
    close(got, ref, tol(dtype, 4), "plain")
    assert int(scratch[:4096].view(torch.int32).abs().sum()) == 0
    full = ops.gemm_splitk(x.to(d), w.to(d), nsplit, bias=bias.to(d), residual=res.to(d), relu=True, scratch=scratch)
    close(full, torch.relu(ref + bias.double()) + res.double(), tol(dtype, 6), "bias + relu + residual")
    for _ in range(3):
        assert torch.equal(full, ops.gemm_splitk(x.to(d), w.to(d), nsplit, bias=bias.to(d), residual=res.to(d), relu=True, scratch=scratch))
    assert int(scratch[:4096].view(torch.int32).abs().sum()) == 0
    one = ops.gemm_nt(x.to(d), w.to(d), bias=bias.to(d), residual=res.to(d), relu=True)
    close(full, one, tol(dtype, 6), "split vs unsplit")
    # LayerNorm folded (packing.ln_folded): each slice sums x and x^2 of its K range, the reducer adds the slices' sums
    g, b = 1 + 0.1 * rnd((K,), 5, torch.float32), 0.1 * rnd((K,), 6, torch.float32)
    lf = packing.ln_folded(None, g, b, bias, "q", dtype, w.float())
    want = cfen_oracle.layer_norm(x.double(), g.double(), b.double()) @ w.double().t() + bias.double()
    gl = ops.gemm_splitk(x.to(d), lf["q.wl"].to(d), nsplit, bias=lf["q.bl"].to(d), lnf_s=lf["q.s"].to(d), scratch=scratch)
    close(gl, want, tol(dtype, 12), "LayerNorm folded")
    assert torch.equal(gl, ops.gemm_splitk(x.to(d), lf["q.wl"].to(d), nsplit, bias=lf["q.bl"].to(d), lnf_s=lf["q.s"].to(d), scratch=scratch))
    close(gl, ops.gemm_ln(x.to(d), lf["q.wl"].to(d), lf["q.s"].to(d), lf["q.bl"].to(d)), tol(dtype, 8), "folded: split vs unsplit")


@pytest.mark.parametrize("release", [1, 0])
def test_gemm_split_k_is_bit_reproducible_with_concurrent_lanes(release):
    """the in-launch split-K reduction under concurrency -- two streams run the GViT-3 shapes (own operands, own scratch) 150 times each while a third
    stream keeps the memory system busy; every result must equal the first one bit for bit and the arrival counters must end at zero.  release = 1: the agent-scope
    release fence in every slice (default, the memory model's recipe); 0: round 3's write-through stores + drain."""
    d = dev()
    ops.tune("gemm.splitk_release", release)
    try:
        lanes = []
        for k, (M, N, K, nsplit) in enumerate([(128, 1536, 6144, 8), (128, 1536, 1536, 4)]):
            x, w = rnd((M, K), 10 + k, torch.float16).to(d), rnd((N, K), 20 + k, torch.float16, 1 / math.sqrt(K)).to(d)
            tiles = ((N + 95) // 96) * ((M + 31) // 32)
            scratch = torch.zeros(4096 + tiles * nsplit * 14336, dtype=torch.uint8, device=d)
            first = ops.gemm_splitk(x, w, nsplit, scratch=scratch).clone()
            lanes.append((x, w, nsplit, scratch, first, torch.cuda.Stream(d), torch.zeros((), dtype=torch.int64, device=d)))
        noise_s, big = torch.cuda.Stream(d), torch.empty(64 << 20, dtype=torch.float32, device=d)
        torch.cuda.synchronize()
        for it in range(150):
            for x, w, nsplit, scratch, first, s, bad in lanes:
                with torch.cuda.stream(s):
                    bad += (ops.gemm_splitk(x, w, nsplit, scratch=scratch) != first).sum()
            with torch.cuda.stream(noise_s):
                big.mul_(1.0001)
        torch.cuda.synchronize()
        for x, w, nsplit, scratch, first, s, bad in lanes:
            assert int(bad) == 0
            assert int(scratch[:4096].view(torch.int32).abs().sum()) == 0
    finally:
        ops.tune("gemm.splitk_release", 1)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kernel", [6])
@pytest.mark.parametrize("M,N,K", [(4096, 1536, 384), (300, 1000, 384), (1000, 1152, 1536), (130, 776, 128), (128, 192, 64)])
def test_gemm_big_tile(dtype, kernel, M, N, K):
    """k_gemm_dma<TM = 4, TN = 6>: 192 x 128 block tile (96 x 64 per wave; a measured-slower variant kept behind "gemm.big"); ragged M
    and N, every epilogue operand, in-place residual"""
    x, w = rnd((M, K), 1, dtype), rnd((N, K), 2, dtype, 1 / math.sqrt(K))
    bias, res, pos = rnd((N,), 3, torch.float32), rnd((M, N), 4, dtype), rnd((16, N), 5, dtype)
    ref = x.double() @ w.double().t()
    d = dev()
    ops.tune("gemm.kernel", kernel)
    try:
        close(ops.gemm_nt(x.to(d), w.to(d)), ref, tol(dtype, 4), "plain")
        want = torch.relu(ref + bias.double()) + res.double() + pos.double()[torch.arange(M) % 16]
        close(ops.gemm_nt(x.to(d), w.to(d), bias=bias.to(d), residual=res.to(d), pos=pos.to(d), relu=True), want, tol(dtype, 8), "bias+relu+res+pos")
        r = res.to(d).clone()
        ops.gemm_nt(x.to(d), w.to(d), residual=r, out=r)
        close(r, ref + res.double(), tol(dtype, 8), "in-place residual")
        base = ops.gemm_nt(x.to(d), w.to(d), bias=bias.to(d), relu=True)
    finally:
        ops.tune("gemm.kernel", -1)
    other = ops.gemm_nt(x.to(d), w.to(d), bias=bias.to(d), relu=True)          # the shape rule's own choice: same sums, same order of K
    assert torch.equal(base, other) or float((base.float() - other.float()).abs().max()) <= tol(dtype, 4)


@pytest.mark.parametrize("dtype", DTYPES)
def test_gemm_big_tile_with_layernorm_folded(dtype):
    from cfen_vit_dehazing_amd.packing import ln_folded
    M, D, N = 2048, 384, 1536
    x = rnd((M, D), 1, dtype, 2.0) + 1.0
    w = rnd((N, D), 2, torch.float32, D ** -0.5)
    g, b, bias = 1 + 0.1 * rnd((D,), 3, torch.float32), 0.1 * rnd((D,), 4, torch.float32), rnd((N,), 5, torch.float32)
    want = (cfen_oracle.layer_norm(x.double(), g.double(), b.double()) @ w.double().t() + bias.double()).relu()
    f = ln_folded(None, g, b, bias, "l", dtype, w)
    ops.tune("gemm.big_min_tiles", 1)
    try:
        ops.tune("gemm.big", 6)
        got = ops.gemm_ln(x.to(dev()), f["l.wl"].to(dev()), f["l.s"].to(dev()), f["l.bl"].to(dev()), relu=True)
        close(got, want, tol(dtype, 8))
    finally:
        ops.tune("gemm.big", 0)
        ops.tune("gemm.big_min_tiles", 256)


def attn_ref(qkv, nseq, S, heads):
    D = qkv.shape[1] // 3
    dh = D // heads
    q, k, v = qkv.double().view(nseq, S, 3, heads, dh).permute(2, 0, 3, 1, 4)
    a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh), -1) @ v
    return a.transpose(1, 2).reshape(nseq * S, D)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("nseq,S,heads,dh", [(3, 256, 4, 24), (2, 16, 4, 24), (2, 4, 8, 24), (2, 1, 16, 24), (2, 64, 2, 96),
                                              (1, 256, 4, 96), (3, 100, 2, 32), (1, 1024, 2, 24), (2, 80, 1, 128)])
def test_attention(dtype, nseq, S, heads, dh):
    qkv = rnd((nseq * S, 3 * heads * dh), 7, dtype, 1.5)
    got = ops.attention(qkv.to(dev()), nseq, S, heads)
    close(got, attn_ref(qkv, nseq, S, heads), tol(dtype, 3))


def to_head_major(qkv, nseq, S, heads):
    """[nseq*S][3D] row-major -> [(seq * heads + head) * 3 + {q,k,v}][S][dh] (what cfen_embed_qkv writes with head_major_heads)"""
    dh = qkv.shape[1] // 3 // heads
    return qkv.view(nseq, S, 3, heads, dh).permute(0, 3, 2, 1, 4).contiguous().view(-1)


@pytest.mark.parametrize("nseq,S,heads", [(3, 256, 4), (5, 256, 8), (2, 64, 4), (9, 64, 16), (2, 1024, 4), (3, 1024, 16)])
def test_attention_head_major_layout(nseq, S, heads):
    qkv = rnd((nseq * S, 3 * heads * 24), 7, torch.float16, 1.5)
    got = ops.attention_head_major(to_head_major(qkv, nseq, S, heads).to(dev()), nseq, S, heads)
    close(got, attn_ref(qkv, nseq, S, heads), tol(torch.float16, 3))
    # a dominating key late in the sequence (the softmax maximum is not among the first keys; S = 1024: not in the first block of 256 keys)
    qkv[900 if S > 256 else 200 % S, heads * 24:heads * 24 + 24] = qkv[3, :24] * 25
    got = ops.attention_head_major(to_head_major(qkv, nseq, S, heads).to(dev()), nseq, S, heads)
    close(got, attn_ref(qkv, nseq, S, heads), tol(torch.float16, 3))
    if S in (256, 1024):   # S = 1024: the 8-wave form of the long-window kernel (key blocks of 256) beside the default 16-wave one (blocks of 128);
        try:               # S = 256: the long-window kernel (two query tiles per K / V fragment) instead of k_attention_hm
            ops.tune("attn.hm_pair", 2 if S == 1024 else 1)
            got8 = ops.attention_head_major(to_head_major(qkv, nseq, S, heads).to(dev()), nseq, S, heads)
        finally:
            ops.tune("attn.hm_pair", 0)
        close(got8, attn_ref(qkv, nseq, S, heads), tol(torch.float16, 3))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_spiky_scores(dtype):
    # one key dominates late in the sequence: exercises the online-softmax rescale
    nseq, S, heads, dh = 1, 256, 1, 24
    qkv = rnd((S, 3 * dh), 11, torch.float32, 0.5)
    qkv[200, dh:2 * dh] = qkv[3, :dh] * 25           # key 200 aligned with query 3
    qkv = qkv.to(dtype)
    close(ops.attention(qkv.to(dev()), nseq, S, heads), attn_ref(qkv, nseq, S, heads), tol(dtype, 3))


# ---------------------------------------------------------------------------------------------------
def perm_tokens(tok_ref, C, p):
    return tok_ref[..., packing.token_perm(C, p)]


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify_roundtrip_lvit(dtype):
    B, C, H, ws = 2, 24, 64, 16
    x = rnd((B, C, H, H), 1, dtype)
    want = perm_tokens(cfen_oracle.unfold_tokens(cfen_oracle.window_partition(x.float(), ws), 2), C, 2).reshape(-1, 4 * C)
    xn = ops.to_nhwc(x).to(dev())
    tok = ops.patchify(xn, C, ws, 2)
    assert torch.equal(tok.float().cpu(), want)
    back = ops.unpatchify(tok, B, H, H, C, xn.shape[-1], ws, 2)
    assert torch.equal(back.cpu(), xn.cpu())


@pytest.mark.parametrize("dtype", DTYPES)
def test_patchify_pooled_gvit_and_padded_stride(dtype):
    B, C, H = 2, 24, 64
    x = rnd((B, C, H, H), 2, dtype)
    pooled = cfen_oracle.avgpool2(cfen_oracle.avgpool2(x.double()))
    want = perm_tokens(cfen_oracle.unfold_tokens(pooled, 4), C, 4).reshape(-1, 16 * C)
    xn = ops.to_nhwc(x, cs=32).to(dev())                       # channel stride larger than C
    tok = ops.patchify(xn, C, H // 4, 4, pool=4)
    close(tok, want, tol(dtype))


@pytest.mark.parametrize("dtype", DTYPES)
def test_upsample4(dtype):
    B, C, h, w = 2, 8, 5, 7
    x = rnd((B, C, h, w), 3, dtype)
    want = cfen_oracle.upsample2_bilinear(cfen_oracle.upsample2_bilinear(x.double()))
    got = ops.from_nhwc(ops.upsample4(ops.to_nhwc(x).to(dev())), C)
    close(got, want, tol(dtype))


def test_tensor2im_u8_matches_reference_vectors(golden_dir):
    """device tensor2im against the vectors captured from the reference's util.tensor2im (edge values, 1-channel tiling)"""
    import numpy as np
    kat = np.load(os.path.join(golden_dir, "ops_kat.npz"))
    for xk, yk in (("t2i/x", "t2i/y"), ("t2i/x3", "t2i/y3")):
        got = ops.tensor2im_u8(torch.from_numpy(kat[xk]).contiguous().to(dev()))
        assert np.array_equal(got.cpu().numpy(), kat[yk])
    x = torch.rand(3, 37, 53) * 2 - 1
    from cfen_vit_dehazing_amd.util import util
    assert np.array_equal(ops.tensor2im_u8(x.to(dev())).cpu().numpy(), util.tensor2im(x))


def test_u8hwc_to_nhwc_equals_totensor_normalize():
    img = torch.randint(0, 256, (2, 24, 40, 3), dtype=torch.uint8)
    want = ((img.permute(0, 3, 1, 2).float() / 255.0) - 0.5) / 0.5          # ToTensor + Normalize(0.5, 0.5)
    for dtype in DTYPES:
        got = ops.u8hwc_to_nhwc(img.to(dev()), 8, dtype)
        assert got.shape == (2, 24, 40, 8) and float(got[..., 3:].abs().max()) == 0.0
        assert torch.equal(got[..., :3].cpu(), want.permute(0, 2, 3, 1).to(dtype))


def test_nchw_to_nhwc():
    x = rnd((2, 3, 16, 32), 4, torch.float32)
    for dtype in DTYPES:
        got = ops.nchw_to_nhwc(x.to(dev()), 8, dtype)
        want = ops.to_nhwc(x, cs=8).to(dtype)
        assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W,ws", [(24, 64, 32, 32), (48, 32, 32, 16), (96, 16, 48, 16)])
def test_embed_gather_equals_patchify_then_gemm(dtype, C, H, W, ws):
    """LViT embedding with the patch gather inside the GEMM loader (k_gemm_nt for K = 96, k_gemm_dma otherwise)"""
    d = dev()
    B, p, D = 3, 2, 4 * C
    S = (ws // p) ** 2
    fmap = ops.to_nhwc(rnd((B, C, H, W), 1, dtype)).to(d)
    w = rnd((D, D), 2, dtype, 1 / math.sqrt(D)).to(d)
    b = rnd((D,), 3, torch.float32, 0.1).to(d)
    pos = rnd((S, D), 4, dtype).to(d)
    tok = ops.patchify(fmap, C, ws, p)
    want = ops.gemm_nt(tok, w, bias=b, residual=tok, pos=pos)
    got = ops.embed_gather(fmap, C, ws, p, w, b, pos)
    assert got.shape == want.shape
    assert torch.equal(got, want)          # same kernels, same operand values, same accumulation order
    ref = tok.double().cpu() @ w.double().cpu().t() + b.double().cpu() + tok.double().cpu() + pos.double().cpu().repeat(tok.shape[0] // S, 1)
    close(got, ref, tol(dtype, 4))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D", [96, 192])
def test_mlp_block_with_projection_prologue(dtype, D):
    """k_mlp with the attention out_proj + residual folded into its token load"""
    d = dev()
    M, H = 300, 4 * D
    x, att = rnd((M, D), 1, dtype), rnd((M, D), 2, dtype)
    wp = rnd((D, D), 3, dtype, 1 / math.sqrt(D))
    w1, w2 = rnd((H, D), 4, dtype, 1 / math.sqrt(D)), rnd((D, H), 5, dtype, 1 / math.sqrt(H))
    b1, b2 = rnd((H,), 6, torch.float32, 0.1), rnd((D,), 7, torch.float32, 0.1)
    g, b = 1 + rnd((D,), 8, torch.float32, 0.1), rnd((D,), 9, torch.float32, 0.1)
    x1 = x.double() + att.double() @ wp.double().t()
    want = x1 + torch.relu(F.layer_norm(x1, (D,), g.double(), b.double(), 1e-5) @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    pk = (lambda w: w[:, packing.kperm32(w.shape[1])].contiguous()) if dtype == torch.float16 else (lambda w: w)
    got = ops.mlp_block(x.to(d), pk(w1).to(d), b1.to(d), pk(w2).to(d), b2.to(d), ln=(g.to(d), b.to(d)), proj=(att.to(d), wp.to(d)))
    close(got, want, tol(dtype, 8))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,H,W,ws", [(24, 64, 32, 32), (48, 32, 48, 16)])
def test_embed_qkv_fused_front(dtype, C, H, W, ws):
    """k_embed_qkv (gather + embedding + residual + pos + LN1 + qkv in one launch) against fp64 and the unfused chain"""
    d = dev()
    B, p, D = 3, 2, 4 * C
    S = (ws // p) ** 2
    fmap = ops.to_nhwc(rnd((B, C, H, W), 1, dtype)).to(d)
    we = rnd((D, D), 2, dtype, 1 / math.sqrt(D)); be = rnd((D,), 3, torch.float32, 0.1)
    pos = rnd((S, D), 4, dtype)
    g, b = 1 + rnd((D,), 5, torch.float32, 0.1), rnd((D,), 6, torch.float32, 0.1)
    wq = rnd((3 * D, D), 7, dtype, 1 / math.sqrt(D))
    perm = packing.kperm32(D) if dtype == torch.float16 else torch.arange(D)
    res = []
    try:
        for lds in (0, 3, 4):               # weights straight from L2 / staged through LDS / LDS-DMA ring (fp16 only): same arithmetic
            ops.tune("embed.lds", lds)
            res.append(ops.embed_qkv(fmap, C, ws, p, we[:, perm].contiguous().to(d), be.to(d), pos.to(d), g.to(d), b.to(d),
                                     wq[:, perm].contiguous().to(d)))
    finally:
        ops.tune("embed.lds", 6)
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    assert torch.equal(res[0][0], res[2][0]) and torch.equal(res[0][1], res[2][1])
    if dtype == torch.float16 and D == 192:
        # the ring depth of k_embed_qkv2 (round 5: 4 stages by default, counted vmcnt waits that leave the younger chunks' DMAs and the tile stores outstanding)
        try:
            ops.tune("embed.lds", 4)
            for ns in (2, 3, 5):
                ops.tune("embed.stages", ns)
                for hm in (0, D // 24):
                    a_, b_ = ops.embed_qkv(fmap, C, ws, p, we[:, perm].contiguous().to(d), be.to(d), pos.to(d), g.to(d), b.to(d), wq[:, perm].contiguous().to(d),
                                           head_major_heads=hm)
                    ops.tune("embed.stages", 4)
                    c_, d_ = ops.embed_qkv(fmap, C, ws, p, we[:, perm].contiguous().to(d), be.to(d), pos.to(d), g.to(d), b.to(d), wq[:, perm].contiguous().to(d),
                                           head_major_heads=hm)
                    ops.tune("embed.stages", ns)
                    assert torch.equal(a_, c_) and torch.equal(b_, d_), "embed.stages %d differs from 4" % ns
        finally:
            ops.tune("embed.stages", 4)
            ops.tune("embed.lds", 6)
    x1, qkv = res[1]
    # head-major store (input layout of cfen_attention_head_major): the same values, laid out per (window, head)
    heads = D // 24
    nwin = B * (H // ws) * (W // ws)
    for lds in (0, 3, 4):
        ops.tune("embed.lds", lds)
        try:
            x1h, qkvh = ops.embed_qkv(fmap, C, ws, p, we[:, perm].contiguous().to(d), be.to(d), pos.to(d), g.to(d), b.to(d),
                                      wq[:, perm].contiguous().to(d), head_major_heads=heads)
        finally:
            ops.tune("embed.lds", 6)
        assert torch.equal(x1h, x1) and torch.equal(qkvh.view(-1), to_head_major(qkv, nwin, S, heads))
    tok = ops.patchify(fmap, C, ws, p)
    t64 = tok.double().cpu()
    y = t64 @ we.double().t() + be.double() + t64 + pos.double().repeat(tok.shape[0] // S, 1)
    close(x1, y, tol(dtype, 4))
    ln = F.layer_norm(y, (D,), g.double(), b.double(), 1e-5)
    close(qkv, ln @ wq.double().t(), tol(dtype, 6))
    # the unfused kernels on the same operands agree to rounding
    y_u = ops.gemm_nt(tok, we.to(d), bias=be.to(d), residual=tok, pos=pos.to(d))
    q_u = ops.gemm_nt(ops.layernorm(y_u, g.to(d), b.to(d)), wq.to(d))
    close(x1, y_u.double(), tol(dtype, 4))
    close(qkv, q_u.double(), tol(dtype, 8))


@pytest.mark.parametrize("H,W,ws,hm", [(64, 32, 32, True), (32, 64, 16, False)])
def test_embed_qkv_stream_front_d384(H, W, ws, hm):
    """k_front3 (row-tile weight streams, one wave per SIMD; C = 96, D = 384, 16 heads of 24) against fp64, row-major and head-major qkv"""
    dtype = torch.float16
    d = dev()
    B, C, p, D, heads = 2, 96, 2, 384, 16
    S = (ws // p) ** 2
    fmap = ops.to_nhwc(rnd((B, C, H, W), 1, dtype)).to(d)
    we, be = rnd((D, D), 2, dtype, D ** -0.5), 0.1 * rnd((D,), 3, torch.float32)
    pos = rnd((S, D), 4, dtype)
    g, b = 1 + 0.1 * rnd((D,), 5, torch.float32), 0.1 * rnd((D,), 6, torch.float32)
    wq = rnd((3 * D, D), 7, dtype, D ** -0.5)
    tok = ops.patchify(fmap, C, ws, p).double().cpu()
    M = tok.shape[0]
    y = tok @ we.double().t() + be.double() + tok + pos.double().repeat(M // S, 1)
    qkv = cfen_oracle.layer_norm(y, g.double(), b.double()) @ wq.double().t()
    kd = packing.kperm32(D)
    x1, got = ops.embed_qkv(fmap, C, ws, p, packing.pack_stream_rows(we[:, kd]).to(d), be.to(d), pos.to(d), g.to(d), b.to(d),
                            packing.pack_stream_rows(wq[:, kd]).to(d), head_major_heads=heads if hm else 0, stream_weights=True)
    close(x1, y, tol(dtype, 6), "x1")
    if hm:
        want = to_head_major(qkv, M // S, S, heads)
        close(got.view(-1), want.reshape(-1), tol(dtype, 12), "qkv head-major")
        att = ops.attention_head_major(got, M // S, S, heads)
        close(att, attn_ref(qkv, M // S, S, heads), tol(dtype, 12), "attention on the streamed front half's qkv")
    else:
        close(got, qkv, tol(dtype, 12), "qkv")
    again = ops.embed_qkv(fmap, C, ws, p, packing.pack_stream_rows(we[:, kd]).to(d), be.to(d), pos.to(d), g.to(d), b.to(d),
                          packing.pack_stream_rows(wq[:, kd]).to(d), head_major_heads=heads if hm else 0, stream_weights=True)
    assert torch.equal(again[0], x1) and torch.equal(again[1], got)


# ---------------------------------------------------------------------------------------------------
def run_conv(dtype, x, w, b, k, stride, pad, reflect=False, an=None, act=0, res=None, nchw=False, x2=None):
    kc = 32 if dtype == torch.float16 else 16
    d = dev()
    cout = w.shape[0]
    cin = x.shape[1]
    wp = packing.pack_conv_weight(w, packing.cs_of(cin), kc, dtype)[0]
    s, t = packing.affine(b, an[0] if an else None, an[1] if an else None, packing.round_up(cout, 16))
    xn = ops.to_nhwc(x.to(dtype)).to(d)
    x2n = ops.to_nhwc(x2.to(dtype)).to(d) if x2 is not None else None
    resn = ops.to_nhwc(res.to(dtype)).to(d) if res is not None else None
    out = ops.conv2d(xn, wp.to(d), s.to(d), t.to(d), packing.cs_of(cin), cout, k=k, stride=stride, pad=pad, reflect=reflect,
                     src1=x2n, act=act, res0=resn, nchw_f32=nchw)
    return out if nchw else ops.from_nhwc(out, cout)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,stride,pad,size", [(3, 12, 5, 1, 2, 32), (12, 12, 3, 1, 1, 32), (12, 24, 3, 2, 1, 64),
                                                       (24, 48, 3, 2, 1, 32), (48, 96, 3, 2, 1, 32)])
def test_conv_plain(dtype, cin, cout, k, stride, pad, size):
    x = rnd((2, cin, size, size), 1, dtype)
    w = rnd((cout, cin, k, k), 2, dtype, 1 / math.sqrt(cin * k * k))
    b = rnd((cout,), 3, torch.float32, 0.1)
    want = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    close(run_conv(dtype, x, w, b, k, stride, pad), want, tol(dtype, 4))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_identity_kernel_is_exact(dtype):
    # asymmetric known answer: a one-hot 3x3 kernel shifts the image (zero padding at the border)
    x = rnd((1, 8, 16, 16), 1, dtype)
    w = torch.zeros(8, 8, 3, 3)
    for c in range(8):
        w[c, (c + 1) % 8, 0, 2] = 1.0                 # out[c](y,x) = in[c+1](y-1, x+1)
    got = run_conv(dtype, x, w.to(dtype), torch.zeros(8), 3, 1, 1)
    want = F.conv2d(x.float(), w, padding=1)
    assert torch.equal(got.float().cpu(), want)


def run_conv_rows(dtype, x, w, b, k, reflect=False, an=None, act=0, res=None, nchw=False):
    """the LDS-tiled kernel behind wlayout=1 (k_conv_tile.hip)"""
    d = dev()
    cout, cin = w.shape[0], x.shape[1]
    cs = packing.cs_of(cin)
    assert packing.conv_uses_rows_layout(dtype, k, 1, k // 2, 1, cs, cout, x.shape[2], x.shape[3])
    wp = packing.pack_conv_weight_rows(w, cs, dtype)[0]
    s, t = packing.affine(b, an[0] if an else None, an[1] if an else None, 16)
    xn = ops.to_nhwc(x.to(dtype)).to(d)
    resn = ops.to_nhwc(res.to(dtype)).to(d) if res is not None else None
    out = ops.conv2d(xn, wp.to(d), s.to(d), t.to(d), cs, cout, k=k, stride=1, pad=k // 2, reflect=reflect, act=act, res0=resn,
                     nchw_f32=nchw, rows_layout=True)
    return out if nchw else ops.from_nhwc(out, cout)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,k,H,W", [(3, 12, 5, 16, 64), (12, 12, 3, 8, 128), (12, 12, 3, 40, 64), (12, 3, 7, 24, 192)])
def test_conv_rows_layout_matches_torch(dtype, cin, cout, k, H, W):
    x = rnd((2, cin, H, W), 1, dtype)
    w = rnd((cout, cin, k, k), 2, dtype, 1 / math.sqrt(cin * k * k))
    b = rnd((cout,), 3, torch.float32, 0.1)
    res = rnd((2, cout, H, W), 4, dtype)
    want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2)) + res.double()
    close(run_conv_rows(dtype, x, w, b, k, act=1, res=res), want, tol(dtype, 4))
    # and bitwise-comparable with the gather kernel on the same operands (same MFMA chunking is not guaranteed: tolerance)
    close(run_conv_rows(dtype, x, w, b, k), run_conv(dtype, x, w, b, k, 1, k // 2).double(), tol(dtype, 4))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_rows_layout_shift_kernel_is_exact(dtype):
    # one-hot taps at the kernel corners: out[c](y,x) = in[c'](y+dy, x+dx) with zero padding, across tile borders
    x = rnd((1, 12, 16, 128), 1, dtype)
    for k in (3, 7):
        w = torch.zeros(12, 12, k, k)
        for c in range(12):
            w[c, (c + 5) % 12, (0 if c % 2 else k - 1), (k - 1 if c % 3 else 0)] = 1.0
        got = run_conv_rows(dtype, x, w.to(dtype), torch.zeros(12), k)
        assert torch.equal(got.float().cpu(), F.conv2d(x.float(), w, padding=k // 2))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_rows_layout_reflect7_tanh_nchw(dtype):
    x = rnd((2, 12, 16, 64), 1, dtype)
    for cout in (3, 1):
        w = rnd((cout, 12, 7, 7), 2, dtype, 0.3 / math.sqrt(12 * 49))
        b = rnd((cout,), 3, torch.float32, 0.1)
        want = torch.tanh(F.conv2d(F.pad(x.double(), (3, 3, 3, 3), mode="reflect"), w.double(), b.double()))
        got = run_conv_rows(dtype, x, w, b, 7, reflect=True, act=2, nchw=True)
        assert got.dtype == torch.float32 and got.shape == want.shape
        close(got, want, tol(dtype, 2))


def test_conv7_toeplitz_reflect_tanh_nchw():
    """k_conv7_tz (4 pixels per MFMA column, Toeplitz weights): tails of the generator, vs torch and vs the rows-layout kernel"""
    dtype, d = torch.float16, dev()
    x = rnd((2, 12, 32, 128), 1, dtype)
    for cout, reflect in ((3, True), (1, True), (4, False)):
        assert packing.conv_uses_toeplitz7(dtype, 7, 1, 3, 1, 16, cout, True, 32, 128)
        w = rnd((cout, 12, 7, 7), 2, dtype, 0.3 / math.sqrt(12 * 49))
        b = rnd((cout,), 3, torch.float32, 0.1)
        xp = F.pad(x.double(), (3, 3, 3, 3), mode="reflect") if reflect else F.pad(x.double(), (3, 3, 3, 3))
        want = torch.tanh(F.conv2d(xp, w.double(), b.double()))
        s, t = packing.affine(b, cout_pad=16)
        got = ops.conv2d(ops.to_nhwc(x).to(d), packing.pack_conv7_toeplitz(w, dtype)[0].to(d), s.to(d), t.to(d), 16, cout, k=7, stride=1, pad=3,
                         reflect=reflect, act=2, nchw_f32=True, toeplitz=True)
        assert got.dtype == torch.float32 and got.shape == want.shape
        close(got, want, tol(dtype, 2))
        close(got, run_conv_rows(dtype, x, w, b, 7, reflect=reflect, act=2, nchw=True).double(), tol(dtype, 2))
    # one-hot taps: pure shifted copies, exact
    w = torch.zeros(3, 12, 7, 7)
    w[0, 5, 0, 6] = 1.0; w[1, 7, 6, 0] = 1.0; w[2, 0, 3, 3] = 1.0
    s, t = packing.affine(torch.zeros(3), cout_pad=16)
    got = ops.conv2d(ops.to_nhwc(x).to(d), packing.pack_conv7_toeplitz(w.to(dtype), dtype)[0].to(d), s.to(d), t.to(d), 16, 3, k=7, stride=1, pad=3,
                     act=0, nchw_f32=True, toeplitz=True)
    assert torch.equal(got.cpu(), F.conv2d(x.float(), w, padding=3))


def test_conv_rows_layout_rejects_unsupported_geometry():
    x = rnd((1, 12, 12, 64), 1, torch.float16)          # H not a multiple of 8
    wp = packing.pack_conv_weight_rows(rnd((12, 12, 3, 3), 2, torch.float16), 16, torch.float16)[0]
    s, t = packing.affine(torch.zeros(12), cout_pad=16)
    with pytest.raises(Exception, match="rows layout"):
        ops.conv2d(ops.to_nhwc(x).to(dev()), wp.to(dev()), s.to(dev()), t.to(dev()), 16, 12, k=3, rows_layout=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_reflect7_tanh_nchw(dtype):
    x = rnd((2, 12, 32, 32), 1, dtype)
    for cout in (3, 1):
        w = rnd((cout, 12, 7, 7), 2, dtype, 0.3 / math.sqrt(12 * 49))
        b = rnd((cout,), 3, torch.float32, 0.1)
        want = torch.tanh(F.conv2d(F.pad(x.double(), (3, 3, 3, 3), mode="reflect"), w.double(), b.double()))
        got = run_conv(dtype, x, w, b, 7, 1, 3, reflect=True, act=2, nchw=True)
        assert got.dtype == torch.float32 and got.shape == want.shape
        close(got, want, tol(dtype, 2))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C", [24, 48, 96])
def test_conv_1x1_concat_actnorm_relu_residual(dtype, C):
    a, b2 = rnd((2, C, 16, 16), 1, dtype), rnd((2, C, 16, 16), 2, dtype)
    w = rnd((C, 2 * C, 1, 1), 3, dtype, 1 / math.sqrt(2 * C))
    bias, anw, anb = rnd((C,), 4, torch.float32, 0.1), rnd((C,), 5, torch.float32, 0.2), rnd((C,), 6, torch.float32, 0.2)
    res = rnd((2, C, 16, 16), 7, dtype)
    y = F.conv2d(torch.cat((a, b2), 1).double(), w.double(), bias.double())
    want = torch.relu((y + anb.double().view(1, -1, 1, 1)) * torch.exp(anw.double()).view(1, -1, 1, 1)) + res.double()
    got = run_conv(dtype, a, w, bias, 1, 1, 0, an=(anw, anb), act=1, res=res, x2=b2)
    close(got, want, tol(dtype, 6))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_gather_with_and_without_lds_staged_weights_is_bitwise_equal(dtype):
    """k_conv reads its weight fragments from an LDS copy staged once per workgroup (default, `conv.wlds` 2: 4- and 8-wave workgroups) or from
    global memory (0): same MFMAs in the same order, so a strided 3x3 conv (96 output channels: the 8-wave shape) and a two-source 1x1 fuse
    conv with ActNorm, ReLU and a residual must agree bit for bit"""
    x = rnd((2, 48, 32, 32), 1, dtype)
    w3, b3 = rnd((96, 48, 3, 3), 2, dtype, 1 / math.sqrt(48 * 9)), rnd((96,), 3, torch.float32, 0.1)
    a, b2, res = rnd((2, 24, 32, 32), 4, dtype), rnd((2, 24, 32, 32), 5, dtype), rnd((2, 24, 32, 32), 6, dtype)
    w1, b1 = rnd((24, 48, 1, 1), 7, dtype, 1 / math.sqrt(48)), rnd((24,), 8, torch.float32, 0.1)
    anw, anb = rnd((24,), 9, torch.float32, 0.2), rnd((24,), 10, torch.float32, 0.2)
    outs = {}
    try:
        for mode in (2, 1, 0):
            ops.tune("conv.wlds", mode)
            outs[mode] = (run_conv(dtype, x, w3, b3, 3, 2, 1, act=1), run_conv(dtype, a, w1, b1, 1, 1, 0, an=(anw, anb), act=1, res=res, x2=b2))
    finally:
        ops.tune("conv.wlds", 2)
    for mode in (1, 0):
        assert torch.equal(outs[mode][0], outs[2][0]) and torch.equal(outs[mode][1], outs[2][1]), mode
    close(outs[2][0], torch.relu(F.conv2d(x.double(), w3.double(), b3.double(), stride=2, padding=1)), tol(dtype, 6))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout", [(96, 48), (48, 24), (24, 12)])
def test_conv_transpose(dtype, cin, cout):
    kc = 32 if dtype == torch.float16 else 16
    d = dev()
    x = rnd((2, cin, 16, 16), 1, dtype)
    w = rnd((cin, cout, 4, 4), 2, dtype, 1 / math.sqrt(cin * 4))
    b = rnd((cout,), 3, torch.float32, 0.1)
    want = torch.relu(F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1))
    wp = packing.pack_convT_weight(w, packing.cs_of(cin), kc, dtype)
    s, t = packing.affine(b, cout_pad=packing.round_up(cout, 16))
    out = ops.conv2d(ops.to_nhwc(x).to(d), wp.to(d), s.to(d), t.to(d), packing.cs_of(cin), cout, transpose=True, act=1)
    close(ops.from_nhwc(out, cout), want, tol(dtype, 4))
    if packing.cs_of(cout) != cout:     # padded channels must be written as zeros
        assert float(out[..., cout:].abs().max()) == 0.0


def test_multi_tile_workgroups_of_conv7_and_convT_cover_ragged_tile_counts():
    """k_conv7_tz walks `conv7.tpw` tiles per workgroup and the 128-byte k_convT_tile `convT.tpw` (weights loaded once per workgroup) when the
    launch has enough tiles; tile counts that do not divide (the last workgroup stops early) must give bitwise the one-tile-per-workgroup result"""
    dtype, d = torch.float16, dev()
    try:
        # conv7: 5 x (1040 / 16) x (576 / 64) = 2925 tiles (odd) -> up to 2 tiles per workgroup (2925 / 1024; the knob value 3 is clamped to 2)
        x = rnd((5, 12, 1040, 576), 1, dtype)
        w = rnd((3, 12, 7, 7), 2, dtype, 0.3 / math.sqrt(12 * 49))
        s, t = packing.affine(rnd((3,), 3, torch.float32, 0.1), cout_pad=16)
        xn, wz = ops.to_nhwc(x).to(d), packing.pack_conv7_toeplitz(w, dtype)[0].to(d)
        outs = {}
        for tpw in (1, 2, 3):
            ops.tune("conv7.tpw", tpw)
            outs[tpw] = ops.conv2d(xn, wz, s.to(d), t.to(d), 16, 3, k=7, stride=1, pad=3, reflect=True, act=2, nchw_f32=True, toeplitz=True).cpu()
        assert torch.equal(outs[1], outs[2]) and torch.equal(outs[1], outs[3])
        last = slice(4, 5)      # the image the ragged last workgroup writes: against torch (affine = scale 1, shift = bias)
        want = torch.tanh(F.conv2d(F.pad(x[last].double(), (3, 3, 3, 3), mode="reflect"), w.double(), t[:3].double()))
        close(outs[2][last], want, tol(dtype, 2))
        # convT (48 -> 24 channels: the 128-byte variant): 3 x (1028 / 4) x (96 / 32) = 2313 tiles (odd) -> 2 tiles per workgroup
        cin, cout = 48, 24
        xt = rnd((3, cin, 1028, 96), 4, dtype)
        wt = rnd((cin, cout, 4, 4), 5, dtype, 1 / math.sqrt(cin * 4))
        st, tt = packing.affine(rnd((cout,), 6, torch.float32, 0.1), cout_pad=packing.round_up(cout, 16))
        xtn, wr = ops.to_nhwc(xt).to(d), packing.pack_convT_weight_rows(wt, packing.cs_of(cin), dtype).to(d)
        outs = {}
        for tpw in (1, 2):
            ops.tune("convT.tpw", tpw)
            outs[tpw] = ops.conv2d(xtn, wr, st.to(d), tt.to(d), packing.cs_of(cin), cout, transpose=True, act=1, rows_layout=True).cpu()
        assert torch.equal(outs[1], outs[2])
    finally:
        ops.tune("conv7.tpw", 4)
        ops.tune("convT.tpw", 2)


@pytest.mark.parametrize("cin,cout,H,W", [(96, 48, 8, 32), (48, 24, 12, 64), (24, 12, 16, 96)])
def test_conv_transpose_rows_layout(cin, cout, H, W):
    """LDS-tiled ConvTranspose2d (k_convT_tile, fp16 only): vs torch, vs the gather kernel, and padded channels zero"""
    dtype, d = torch.float16, dev()
    assert packing.convT_uses_rows_layout(dtype, packing.cs_of(cin), cout, H, W)
    x = rnd((2, cin, H, W), 1, dtype)
    w = rnd((cin, cout, 4, 4), 2, dtype, 1 / math.sqrt(cin * 4))
    b = rnd((cout,), 3, torch.float32, 0.1)
    anw, anb = rnd((cout,), 4, torch.float32, 0.2), rnd((cout,), 5, torch.float32, 0.2)
    y = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2, padding=1)
    want = torch.relu((y + anb.double().view(1, -1, 1, 1)) * torch.exp(anw.double()).view(1, -1, 1, 1))
    s, t = packing.affine(b, anw, anb, packing.round_up(cout, 16))
    xn = ops.to_nhwc(x).to(d)
    out = ops.conv2d(xn, packing.pack_convT_weight_rows(w, packing.cs_of(cin), dtype).to(d), s.to(d), t.to(d), packing.cs_of(cin), cout,
                     transpose=True, act=1, rows_layout=True)
    close(ops.from_nhwc(out, cout), want, tol(dtype, 4))
    ref = ops.conv2d(xn, packing.pack_convT_weight(w, packing.cs_of(cin), 32, dtype).to(d), s.to(d), t.to(d), packing.cs_of(cin), cout,
                     transpose=True, act=1)
    close(ops.from_nhwc(out, cout), ops.from_nhwc(ref, cout).double(), tol(dtype, 4))
    if packing.cs_of(cout) != cout:
        assert float(out[..., cout:].abs().max()) == 0.0


def test_conv_transpose_rows_layout_one_hot_is_exact():
    # each output parity picks exactly one kernel tap per input pixel: a one-hot weight makes the output a (shifted) copy
    dtype, d = torch.float16, dev()
    x = rnd((1, 24, 8, 64), 1, dtype)
    w = torch.zeros(24, 12, 4, 4)
    for co in range(12):
        w[(2 * co + 1) % 24, co, co % 4, (co // 4 + 1) % 4] = 1.0
    want = F.conv_transpose2d(x.float(), w, stride=2, padding=1)
    s, t = packing.affine(torch.zeros(12), cout_pad=16)
    out = ops.conv2d(ops.to_nhwc(x).to(d), packing.pack_convT_weight_rows(w.to(dtype), 24, dtype).to(d), s.to(d), t.to(d), 24, 12,
                     transpose=True, rows_layout=True)
    assert torch.equal(ops.from_nhwc(out, 12).float().cpu(), want)


@pytest.mark.parametrize("dtype", DTYPES)
def test_instnorm_relu(dtype):
    for C, size in ((24, 32), (96, 16), (48, 8)):
        x = rnd((2, C, size, size), 1, dtype, 2.0) + 0.7
        want = torch.relu(cfen_oracle.instance_norm(x.double()))
        xn = ops.to_nhwc(x).to(dev())
        close(ops.from_nhwc(ops.instnorm_relu_(xn, C), C), want, tol(dtype, 2))


@pytest.mark.parametrize("dtype", DTYPES)
def test_cfsm2g_against_reference_vectors(dtype, golden_dir):
    kat = np.load(golden_dir + "/ops_kat.npz")
    sd = {"c." + k[len("cfsm/sd/"):]: torch.from_numpy(kat[k]) for k in kat.files if k.startswith("cfsm/sd/")}
    xs = [torch.from_numpy(kat["cfsm/x%d" % i]) for i in range(3)]
    w = torch.cat([sd["c.%s.%d.weight" % (fc, i)].reshape(-1) for fc in ("fc_avg_cf1", "fc_avg_cf2", "fc_max_cf1", "fc_max_cf2")
                   for i in (0, 2)])
    d = dev()
    xn = [ops.to_nhwc(x.to(dtype)).to(d) for x in xs]
    got = ops.from_nhwc(ops.cfsm2g(xn[0], xn[1], xn[2], w.to(d), 8), 8)
    close(got, torch.from_numpy(kat["cfsm/y"]), tol(dtype, 4))


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("D,H,M", [(96, 384, 300), (96, 192, 64), (192, 768, 200), (192, 384, 130), (96, 128, 1000)])
def test_fused_mlp_block(dtype, D, H, M):
    x = rnd((M, D), 1, dtype)
    g, b = 1 + 0.1 * rnd((D,), 2, torch.float32), 0.1 * rnd((D,), 3, torch.float32)
    w1a, w2a = rnd((H, D), 4, dtype, D ** -0.5), rnd((D, H), 5, dtype, 0.5 * H ** -0.5)
    w1b, w2b = rnd((H, D), 6, dtype, D ** -0.5), rnd((D, H), 7, dtype, 0.5 * H ** -0.5)
    b1a, b2a, b1b, b2b = (0.1 * rnd((n,), 8 + i, torch.float32) for i, n in enumerate((H, D, H, D)))
    xd = x.double()
    y1 = xd + torch.relu(cfen_oracle.layer_norm(xd, g.double(), b.double()) @ w1a.double().t() + b1a.double()) @ w2a.double().t() + b2a.double()
    y2 = y1 + torch.relu(y1 @ w1b.double().t() + b1b.double()) @ w2b.double().t() + b2b.double()
    d = dev()
    if dtype == torch.float16:
        kd, kh = packing.kperm32(D), packing.kperm32(H)
        p = lambda w1, w2: (w1[:, kd].contiguous().to(d), w2[:, kh].contiguous().to(d))
    else:
        p = lambda w1, w2: (w1.to(d), w2.to(d))
    wa, wb = p(w1a, w2a), p(w1b, w2b)
    one = ops.mlp_block(x.to(d), wa[0], b1a.to(d), wa[1], b2a.to(d), ln=(g.to(d), b.to(d)))
    close(one, y1, tol(dtype, 6), "stage a")
    two = ops.mlp_block(x.to(d), wa[0], b1a.to(d), wa[1], b2a.to(d), ln=(g.to(d), b.to(d)), second=(wb[0], b1b.to(d), wb[1], b2b.to(d)))
    close(two, y2, tol(dtype, 10), "both stages")
    nol = ops.mlp_block(x.to(d), wa[0], b1a.to(d), wa[1], b2a.to(d))
    close(nol, xd + torch.relu(xd @ w1a.double().t() + b1a.double()) @ w2a.double().t() + b2a.double(), tol(dtype, 6), "no LN")


@pytest.mark.parametrize("dtype", DTYPES)
def test_fused_mlp_fold_epilogue(dtype):
    # D = 96 = 2*2*24: 8x8 windows of 2x2 patches on a 16x32 map, padded channel stride
    B, C, Hm, Wm, ws, pp, cs = 2, 24, 16, 32, 8, 2, 32
    D, H = 96, 192
    M = B * Hm * Wm // 4
    x = rnd((M, D), 1, dtype)
    w1, w2 = rnd((H, D), 2, dtype, D ** -0.5), rnd((D, H), 3, dtype, H ** -0.5)
    b1, b2 = 0.1 * rnd((H,), 4, torch.float32), 0.1 * rnd((D,), 5, torch.float32)
    d = dev()
    if dtype == torch.float16:
        w1p, w2p = w1[:, packing.kperm32(D)].contiguous(), w2[:, packing.kperm32(H)].contiguous()
    else:
        w1p, w2p = w1, w2
    tok = ops.mlp_block(x.to(d), w1p.to(d), b1.to(d), w2p.to(d), b2.to(d))
    fm = ops.mlp_block(x.to(d), w1p.to(d), b1.to(d), w2p.to(d), b2.to(d), fold=(B, Hm, Wm, C, cs, ws, pp))
    want = ops.unpatchify(tok, B, Hm, Wm, C, cs, ws, pp)
    assert torch.equal(fm, want)


@pytest.mark.parametrize("D,H,M", [(384, 1536, 300), (384, 768, 128), (384, 96, 31), (192, 768, 520), (192, 384, 256)])
def test_mlp_stream_block(D, H, M):
    """k_mlp3 (fragment-stream weights, one wave per SIMD): out_proj prologue + LN + FFN + mlp_head against fp64, every optional part on and off,
    ragged token counts; the fold epilogue against unpatchify of the token-major result"""
    dtype = torch.float16
    d = dev()
    x, att = rnd((M, D), 1, dtype), rnd((M, D), 2, dtype)
    wp = rnd((D, D), 3, dtype, D ** -0.5)
    g, b = 1 + 0.1 * rnd((D,), 4, torch.float32), 0.1 * rnd((D,), 5, torch.float32)
    w1a, w2a = rnd((H, D), 6, dtype, D ** -0.5), rnd((D, H), 7, dtype, 0.5 * H ** -0.5)
    w1b, w2b = rnd((H, D), 8, dtype, D ** -0.5), rnd((D, H), 9, dtype, 0.5 * H ** -0.5)
    b1a, b2a, b1b, b2b = (0.1 * rnd((n,), 10 + i, torch.float32) for i, n in enumerate((H, D, H, D)))
    kd, kh = packing.kperm32(D), packing.kperm32(H)
    sa = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]).to(d)
    sb = packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]).to(d)
    sp = packing.pack_stream_sq(wp).to(d)
    xd = x.double()
    ffn = lambda v, w1, b1, w2, b2, ln: v + torch.relu((cfen_oracle.layer_norm(v, g.double(), b.double()) if ln else v) @ w1.double().t()
                                                       + b1.double()) @ w2.double().t() + b2.double()
    y1 = ffn(xd, w1a, b1a, w2a, b2a, True)
    close(ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d))), y1, tol(dtype, 6), "stage a")
    close(ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H), ffn(xd, w1a, b1a, w2a, b2a, False), tol(dtype, 6), "no LN")
    y2 = ffn(y1, w1b, b1b, w2b, b2b, False)
    two = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)))
    close(two, y2, tol(dtype, 10), "both stages")
    x1 = xd + att.double() @ wp.double().t()
    full = ffn(ffn(x1, w1a, b1a, w2a, b2a, True), w1b, b1b, w2b, b2b, False)
    got = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)), proj=(att.to(d), sp))
    close(got, full, tol(dtype, 12), "projection prologue + both stages")
    again = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)), proj=(att.to(d), sp))
    assert torch.equal(got, again)
    if D == 192:
        # the workgroup shapes of the D = 192 variant (default 22: two 78 KB workgroups a CU on a three-slot ring, 256 registers; 24: four slots; 2 / 3 / 4 token
        # tiles a wave on the six-slot ring of one workgroup a CU) do the same arithmetic per token in the same order
        try:
            for tm in (24, 3, 4, 2):
                ops.tune("mlp3.tm192", tm)
                other = ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)), proj=(att.to(d), sp))
                if tm == 2:     # (the two-tile shape on 512 registers: hipcc contracts one epilogue product differently -- 1 fp16 ulp on a handful of elements)
                    assert float((got.float() - other.float()).abs().max()) <= 2e-3
                else:
                    assert torch.equal(got, other), "mlp3.tm192 = %d differs from the default shape" % tm
        finally:
            ops.tune("mlp3.tm192", 22)


def test_mlp_stream_fold_epilogue():
    # D = 384 = 2*2*96: 8x8 windows of 2x2 patches on a 16x32 map
    dtype = torch.float16
    B, C, Hm, Wm, ws, pp, cs = 2, 96, 16, 32, 8, 2, 96
    D, H = 384, 256
    M = B * Hm * Wm // 4
    x = rnd((M, D), 1, dtype)
    w1, w2 = rnd((H, D), 2, dtype, D ** -0.5), rnd((D, H), 3, dtype, H ** -0.5)
    b1, b2 = 0.1 * rnd((H,), 4, torch.float32), 0.1 * rnd((D,), 5, torch.float32)
    d = dev()
    st = packing.pack_stream_pair(w1[:, packing.kperm32(D)], w2[:, packing.kperm32(H)]).to(d)
    tok = ops.mlp_stream_block(x.to(d), st, b1.to(d), b2.to(d), H)
    fm = ops.mlp_stream_block(x.to(d), st, b1.to(d), b2.to(d), H, fold=(B, Hm, Wm, C, cs, ws, pp))
    assert torch.equal(fm, ops.unpatchify(tok, B, Hm, Wm, C, cs, ws, pp))


@pytest.mark.parametrize("H,M,fold", [(1536, 24 * 128 + 77, False), (512, 256, True), (256, 130, False)])
def test_mlp_stream_pair_kernel(H, M, fold):
    """k_mlp3p (the D = 384 block on wave pairs, "mlp3.pair" = 1, the default): against fp64, against k_mlp3<24, ...> on the same operands (the LayerNorm sums associate over
    the pair: 1-2 fp16 ulp, not bit-equal), run to run, ragged token counts, the fold epilogue against unpatchify of its own token-major result"""
    dtype = torch.float16
    d = dev()
    D = 384
    B, C, Hm, Wm, ws, pp, cs = 2, 96, 16, 32, 8, 2, 96
    x, att = rnd((M, D), 1, dtype), rnd((M, D), 2, dtype)
    wp = rnd((D, D), 3, dtype, D ** -0.5)
    g, b = 1 + 0.1 * rnd((D,), 4, torch.float32), 0.1 * rnd((D,), 5, torch.float32)
    w1a, w2a = rnd((H, D), 6, dtype, D ** -0.5), rnd((D, H), 7, dtype, 0.5 * H ** -0.5)
    w1b, w2b = rnd((H, D), 8, dtype, D ** -0.5), rnd((D, H), 9, dtype, 0.5 * H ** -0.5)
    b1a, b2a, b1b, b2b = (0.1 * rnd((n,), 10 + i, torch.float32) for i, n in enumerate((H, D, H, D)))
    kd, kh = packing.kperm32(D), packing.kperm32(H)
    sa = packing.pack_stream_pair(w1a[:, kd], w2a[:, kh]).to(d)
    sb = packing.pack_stream_pair(w1b[:, kd], w2b[:, kh]).to(d)
    sp = packing.pack_stream_sq(wp).to(d)
    ffn = lambda v, w1, b1, w2, b2, ln: v + torch.relu((cfen_oracle.layer_norm(v, g.double(), b.double()) if ln else v) @ w1.double().t()
                                                       + b1.double()) @ w2.double().t() + b2.double()
    full = ffn(ffn(x.double() + att.double() @ wp.double().t(), w1a, b1a, w2a, b2a, True), w1b, b1b, w2b, b2b, False)
    call = lambda **kw: ops.mlp_stream_block(x.to(d), sa, b1a.to(d), b2a.to(d), H, ln=(g.to(d), b.to(d)), second=(sb, b1b.to(d), b2b.to(d)), proj=(att.to(d), sp), **kw)
    try:
        ops.tune("mlp3.pair", 1)
        got = call()
        close(got, full, tol(dtype, 12), "pair kernel")
        assert torch.equal(got, call())
        if fold:
            assert torch.equal(call(fold=(B, Hm, Wm, C, cs, ws, pp)), ops.unpatchify(got, B, Hm, Wm, C, cs, ws, pp))
        ops.tune("mlp3.pair", 0)
        single = call()
        assert float((got.float() - single.float()).abs().max()) <= 2 ** -7 * max(1.0, float(single.float().abs().max()) / 4)
    finally:
        ops.tune("mlp3.pair", 1)


# ---------------------------------------------------------------------------------------------------
def _lvit_instance(seed):
    """one LViT level-1 instance (C = 24, D = 96, 4 heads, hidden 384) with the deterministic 'trained' weight distribution"""
    from cfen_vit_dehazing_amd.config import NetConfig
    from cfen_vit_dehazing_amd.manifest import generate_state_dict
    cfg = NetConfig(24, 4, patch_size=32, load_size=256)
    g = cfg.vit("localvit_encoder_01")
    full = generate_state_dict(cfg, seed=seed, with_dead=False)
    sd = {k: v for k, v in full.items() if k.startswith(g.name + ".")}
    return cfg, g, sd


@pytest.mark.parametrize("B,H,W", [(1, 32, 32), (2, 64, 96)])
def test_lvit_window_block_against_oracle_and_unfused_chain(B, H, W):
    """k_lvit_window: the whole LViT block map -> map in one launch (q / k / v / attention output never leave the chip) vs the fp64
    oracle on the fp16-rounded weights, and vs the shipped three-kernel chain on the same operands"""
    cfg, g, sd = _lvit_instance(3)
    d = dev()
    dt = torch.float16
    sd16 = {k: (v.to(dt) if v.dtype.is_floating_point else v) for k, v in sd.items()}
    x = rnd((B, 24, H, W), 5, dt)
    want = cfen_oracle.lvit({k: (v.double() if v.dtype.is_floating_point else v) for k, v in sd16.items()}, g.name, x.double(), g.heads, 32)
    pk = packing.pack_vit(sd16, g, dt)
    pk.update(packing.pack_lvit_window(sd16, g, dt))
    pk = {k: v.to(d).contiguous() for k, v in pk.items()}
    fmap = ops.to_nhwc(x).to(d)
    got = ops.from_nhwc(ops.lvit_window(fmap, 24, 32, 2, pk, g.name, g.hidden), 24)
    close(got, want, tol(dt, 12), "fused window block vs fp64")
    # the three workgroup shapes (16 waves x 1 token tile: the default; 8 x 2; 4 x 4 on 512 registers) do the same arithmetic per token
    try:
        for shape in (0, 1, 12, 15, 4, 6):       # 12: the refill right behind the barrier; 15: double MLP chunks over the dead K / V tiles (round 6)          # 6: the embedding / K, V matrices in 64-row chunks (5 front chunks instead of 9)          # 4: the attention loops' K / V fragment reads issued by hand ahead of the MFMAs (round 5): same arithmetic, same order
            ops.tune("lvit.shape", shape)
            other = ops.from_nhwc(ops.lvit_window(fmap, 24, 32, 2, pk, g.name, g.hidden), 24)
            assert torch.equal(other, got), "lvit.shape %d differs from the default shape" % shape
    finally:
        ops.tune("lvit.shape", 2)
    # the unfused chain (embed_qkv -> attention -> mlp with projection prologue + fold)
    n = g.name
    x1, qkv = ops.embed_qkv(fmap, 24, 32, 2, pk[n + ".embed.wk"], pk[n + ".embed.b"], pk[n + ".pos"], pk[n + ".ln1.g"], pk[n + ".ln1.b"],
                            pk[n + ".qkv.wk"], head_major_heads=g.heads)
    nwin = B * (H // 32) * (W // 32)
    att = ops.attention_head_major(qkv, nwin, 256, g.heads)
    chain = ops.mlp_block(x1, pk[n + ".ffn1.wk"], pk[n + ".ffn1.b"], pk[n + ".ffn2.wk"], pk[n + ".ffn2.b"], ln=(pk[n + ".ln2.g"], pk[n + ".ln2.b"]),
                          second=(pk[n + ".head1.wk"], pk[n + ".head1.b"], pk[n + ".head2.wk"], pk[n + ".head2.b"]),
                          proj=(att, pk[n + ".proj.w"]), fold=(B, H, W, 24, 24, 32, 2))
    close(ops.from_nhwc(chain, 24), want, tol(dt, 12), "unfused chain vs fp64")
    assert float((got.float() - ops.from_nhwc(chain, 24).float()).abs().max()) <= tol(dt, 12)


# ---------------------------------------------------------------------------------------------------
def _chain_ref(x, w, bias=None, lnf=None, residual=None, pos=None, relu=False):
    """one phase in float64 from the fp16-rounded operands (the kernel stores fp16 between phases: the caller rounds)"""
    y = x.double() @ w.double().t()
    if lnf is not None:
        g, b = lnf
        y = cfen_oracle.layer_norm(x.double(), g.double(), b.double()) @ w.double().t()
    if bias is not None:
        y = y + bias.double()
    if relu:
        y = torch.relu(y)
    if residual is not None:
        y = y + residual.double()
    if pos is not None:
        y = y + pos.double()[torch.arange(x.shape[0]) % pos.shape[0]]
    return y


@pytest.mark.parametrize("M,N,K,nsplit,team", [(128, 1536, 1536, 4, 48), (128, 1536, 6144, 4, 48), (512, 768, 3072, 2, 48), (2048, 384, 1536, 1, 48),
                                             (16, 128, 64, 1, 3), (100, 256, 512, 2, 5), (256, 1152, 384, 1, 7), (130, 128, 256, 4, 64)])
def test_gemm_chain_single_phase(M, N, K, nsplit, team):
    """k_gvit_chain, one phase: every epilogue operand, ragged token counts (clamped rows, skipped stores), split-K with the in-launch
    reduction, in-place residual; bit-reproducible; against float64"""
    d = dev()
    dt = torch.float16
    x, w = rnd((M, K), 1, dt), rnd((N, K), 2, dt, 1 / math.sqrt(K))
    bias, res, pos = rnd((N,), 3, torch.float32), rnd((M, N), 4, dt), rnd((16, N), 5, dt)
    y = torch.full((M, N), float("nan"), dtype=dt, device=d)
    assert ops.gemm_chain([dict(x=x.to(d), w=w.to(d), y=y, nsplit=nsplit)], M, team) == 0
    close(y, _chain_ref(x, w), tol(dt, 4), "plain")
    y2 = torch.empty_like(y)
    ph = dict(x=x.to(d), w=w.to(d), y=y2, bias=bias.to(d), residual=res.to(d), pos=pos.to(d), relu=True, nsplit=nsplit)
    assert ops.gemm_chain([ph], M, team) == 0
    close(y2, _chain_ref(x, w, bias, None, res, pos, True), tol(dt, 8), "bias + relu + residual + pos")
    y3 = torch.empty_like(y)
    ph["y"] = y3
    for _ in range(3):
        assert ops.gemm_chain([ph], M, team) == 0
        assert torch.equal(y2, y3)
    r = res.to(d).clone()     # in-place residual (proj / linear2 of the block)
    assert ops.gemm_chain([dict(x=x.to(d), w=w.to(d), y=r, residual=r, bias=bias.to(d), nsplit=nsplit)], M, team) == 0
    close(r, _chain_ref(x, w, bias, None, res), tol(dt, 8), "in-place residual")
    if nsplit == 1:
        g, b = 1 + 0.1 * rnd((K,), 6, torch.float32), 0.1 * rnd((K,), 7, torch.float32)
        lf = packing.ln_folded(None, g, b, bias, "q", dt, w.float())
        y4 = torch.empty_like(y)
        assert ops.gemm_chain([dict(x=x.to(d), w=lf["q.wl"].to(d), y=y4, bias=lf["q.bl"].to(d), lnf_s=lf["q.s"].to(d), relu=True)], M, team) == 0
        close(y4, _chain_ref(x, w, bias, (g, b), relu=True), tol(dt, 12), "LayerNorm folded")


@pytest.mark.parametrize("B,S,D,H,heads,team", [(8, 16, 1536, 6144, 16, 48), (8, 64, 768, 3072, 8, 48), (2, 256, 384, 1536, 4, 48), (8, 64, 768, 768, 8, 24),
                                               (1, 16, 384, 768, 4, 9)])
def test_gemm_chain_gvit_block(B, S, D, H, heads, team):
    """the two persistent chains of a GViT instance (embed -> qkv; proj -> ffn1 -> ffn2 -> head1 -> head2 + fold) around the attention operator,
    against a float64 restatement of v3:1272-1325 / 1359-1390 on the same fp16-rounded parameters; the phases hand their outputs to each other
    through the grid barrier (in-place x1, shared hidden buffer)"""
    d, dt = dev(), torch.float16
    M = B * S
    C = D // 16
    tw = int(math.isqrt(S))
    mapH = 4 * tw
    sc = lambda k: 1 / math.sqrt(k)
    x0 = rnd((M, D), 1, dt)
    we, be, pos = rnd((D, D), 2, dt, sc(D)), rnd((D,), 3, torch.float32, 0.1), rnd((S, D), 4, dt, 0.5)
    g1, b1, g2, b2 = [(1 if i % 2 == 0 else 0) + 0.1 * rnd((D,), 5 + i, torch.float32) for i in range(4)]
    wqkv, wp = rnd((3 * D, D), 9, dt, sc(D)), rnd((D, D), 10, dt, sc(D))
    w1, bb1, w2, bb2 = rnd((H, D), 11, dt, sc(D)), rnd((H,), 12, torch.float32, 0.1), rnd((D, H), 13, dt, sc(H)), rnd((D,), 14, torch.float32, 0.1)
    w3, bb3, w4, bb4 = rnd((H, D), 15, dt, sc(D)), rnd((H,), 16, torch.float32, 0.1), rnd((D, H), 17, dt, sc(H)), rnd((D,), 18, torch.float32, 0.1)
    lq = packing.ln_folded(None, g1, b1, None, "q", dt, wqkv.float())
    lf = packing.ln_folded(None, g2, b2, bb1, "f", dt, w1.float())
    T = lambda t: t.to(d)
    x1 = torch.empty(M, D, dtype=dt, device=d)
    qkv = torch.empty(M, 3 * D, dtype=dt, device=d)
    hid = torch.empty(M, H, dtype=dt, device=d)
    sm = torch.zeros(B, mapH, mapH, C, dtype=dt, device=d)
    x0d = T(x0)

    def nsp(N, K):
        n, units = 1, ((M + 127) // 128) * (N // 128)
        while units * n * 2 <= team and n < 8 and (K // 64) % (2 * n) == 0 and K // 64 // (2 * n) >= 4:
            n *= 2
        return n

    err = ops.gemm_chain([dict(x=x0d, w=T(we), y=x1, bias=T(be), residual=x0d, pos=T(pos), nsplit=nsp(D, D)),
                          dict(x=x1, w=T(lq["q.wl"]), y=qkv, bias=T(lq["q.bl"]), lnf_s=T(lq["q.s"]))], M, team)
    assert err == 0
    # float64 reference, rounding to fp16 where the kernels store fp16
    r16 = lambda t: t.to(dt).double()
    X1 = r16(_chain_ref(x0, we, be, None, x0, pos))
    close(x1, X1, tol(dt, 8), "embed")
    QKV = r16(cfen_oracle.layer_norm(X1, g1.double(), b1.double()) @ wqkv.double().t())
    close(qkv, QKV, tol(dt, 16), "qkv")
    att = ops.attention(qkv, B, S, heads)
    q, k, v = [QKV[:, i * D:(i + 1) * D].reshape(B, S, heads, D // heads).permute(0, 2, 1, 3) for i in range(3)]
    A = r16((torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(D // heads), -1) @ v).permute(0, 2, 1, 3).reshape(M, D))
    close(att, A, tol(dt, 8), "attention")
    err = ops.gemm_chain([dict(x=att, w=T(wp), y=x1, residual=x1, nsplit=nsp(D, D)),
                          dict(x=x1, w=T(lf["f.wl"]), y=hid, bias=T(lf["f.bl"]), lnf_s=T(lf["f.s"]), relu=True),
                          dict(x=hid, w=T(w2), y=x1, bias=T(bb2), residual=x1, nsplit=nsp(D, H)),
                          dict(x=x1, w=T(w3), y=hid, bias=T(bb3), relu=True),
                          dict(x=hid, w=T(w4), y=sm, bias=T(bb4), residual=x1, nsplit=nsp(D, H), fold=True)], M, team, fold=(mapH, mapH, C, C, 4))
    assert err == 0
    X2 = r16(att.double().cpu() @ wp.double().t() + X1)
    HID = r16(torch.relu(cfen_oracle.layer_norm(X2, g2.double(), b2.double()) @ w1.double().t() + bb1.double()))
    X3 = r16(HID @ w2.double().t() + bb2.double() + X2)
    HID2 = r16(torch.relu(X3 @ w3.double().t() + bb3.double()))
    Y = HID2 @ w4.double().t() + bb4.double() + X3          # [M][D], feature (i, j, c)
    want = Y.reshape(B, tw, tw, 4, 4, C).permute(0, 1, 3, 2, 4, 5).reshape(B, mapH, mapH, C)
    close(sm, want, tol(dt, 40), "block output (folded map)")


@pytest.mark.parametrize("u8", [False, True])
@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (1, 8, 64), (3, 24, 192)])
def test_head_conv5_from_the_network_input(u8, B, H, W):
    """k_head5: head.0.0 (5x5, 3 -> 12, zero padding 2, bias) straight from the fp32 NCHW input / the uint8 HWC image (ToTensor + Normalize(0.5, 0.5)
    folded in) against F.conv2d on the fp16-rounded operands; borders, several tiles, padded channels exactly zero"""
    d = dev()
    w, bias = rnd((12, 3, 5, 5), 1, torch.float16, 0.2), rnd((12,), 2, torch.float32)
    if u8:
        img = torch.randint(0, 256, (B, H, W, 3), generator=torch.Generator().manual_seed(3), dtype=torch.uint8)
        x = ((img.float() / 255 - 0.5) / 0.5).permute(0, 3, 1, 2)
        got = ops.head_conv5(img.to(d), w.to(d), bias.to(d))
    else:
        x = torch.rand(B, 3, H, W, generator=torch.Generator().manual_seed(3)) * 2 - 1
        got = ops.head_conv5(x.to(d).contiguous(), w.to(d), bias.to(d))
    want = F.conv2d(x.half().double(), w.double(), bias.double(), padding=2).permute(0, 2, 3, 1)
    close(got[..., :12], want, tol(torch.float16, 4), "conv5 from the input")
    assert float(got[..., 12:].abs().max()) == 0.0
    relu = ops.head_conv5((img if u8 else x).to(d).contiguous(), w.to(d), bias.to(d), act=1)
    close(relu[..., :12], torch.relu(want), tol(torch.float16, 4), "with ReLU")
