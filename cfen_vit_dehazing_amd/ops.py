"""Python handles on the individual HIP operators of libcfen_hip.so (tensors in, tensors out).

Used by the parity tests and by anyone who wants one fused block instead of the whole generator.
Every function launches on torch's current CUDA(=HIP) stream and raises CfenError on failure; there
is no PyTorch fallback.
"""
import ctypes

import torch

from . import _lib
from ._lib import ptr, check, dtype_code, current_stream, ConvArgsC, MlpArgsC
from .packing import round_up


def _cuda(*ts):
    for t in ts:
        if t is not None and (not t.is_cuda or not t.is_contiguous()):
            raise ValueError("HIP operators need contiguous CUDA tensors")


_TUNED = {}


def tune(key, value):
    """process-wide kernel-variant knob (cfen_tune); for benchmarks"""
    check(_lib.load().cfen_tune(key.encode(), int(value)), "tune")
    _TUNED[key] = int(value)


def tuned(key, default=None):
    """what this process last set a knob to through tune() (None / `default` = never touched: the library's own default is in force)"""
    return _TUNED.get(key, default)


def gemm_nt(x, w, bias=None, residual=None, pos=None, relu=False, out=None):
    """act(x @ w.T + bias) + residual + pos[row % len(pos)]   (x: [M,K], w: [N,K])"""
    _cuda(x, w, bias, residual, pos, out)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    lib = _lib.load()
    check(lib.cfen_gemm_nt(dtype_code(x.dtype), ptr(x), K, ptr(w), K, ptr(bias), ptr(residual), N, ptr(pos),
                           pos.shape[0] if pos is not None else 0, ptr(out), N, M, N, K, int(relu), current_stream()), "gemm_nt")
    return out


def gemm_ln(x, wl, s, bias=None, relu=False, eps=1e-5):
    """act(LayerNorm-statistics(x) applied to x @ wl.T: rstd (x wl^T - mean s) + bias); wl / s / bias from packing.ln_folded"""
    _cuda(x, wl, s, bias)
    M, K = x.shape
    N = wl.shape[0]
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    check(_lib.load().cfen_gemm_ln(dtype_code(x.dtype), ptr(x), K, ptr(wl), K, ptr(s), ptr(bias), ptr(out), N, M, N, K, int(relu), eps,
                                   current_stream()), "gemm_ln")
    return out


def gemm_splitk(x, w, nsplit, bias=None, residual=None, relu=False, lnf_s=None, scratch=None):
    """cfen_gemm_splitk: act(x @ w.T + bias) + residual (or the LayerNorm-folded form when lnf_s is given) with K cut into nsplit slices and
    the in-launch reduction; `scratch` (zeroed uint8 buffer) can be passed to check that calls leave its counters zero"""
    _cuda(x, w, bias, residual, lnf_s, scratch)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty(M, N, dtype=x.dtype, device=x.device)
    tiles = ((N + 95) // 96) * ((M + 31) // 32)
    if scratch is None:
        scratch = torch.zeros(4096 + tiles * nsplit * 14336, dtype=torch.uint8, device=x.device)
    check(_lib.load().cfen_gemm_splitk(dtype_code(x.dtype), ptr(x), K, ptr(w), K, ptr(lnf_s), ptr(bias), ptr(residual), N, ptr(out), N, M, N, K,
                                       int(relu), nsplit, ptr(scratch), scratch.numel(), current_stream()), "gemm_splitk")
    return out


def head_conv5(x, w, bias, act=0):
    """cfen_head_conv5: conv5x5 (pad 2) of the network input -- (B,3,H,W) fp32 NCHW or (B,H,W,3) uint8 -- to a (B,H,W,16) fp16 NHWC map;
    w: (Cout <= 16, 3, 5, 5), bias: (Cout,)"""
    from .packing import pack_head5
    _cuda(x, w, bias)
    u8 = x.dtype == torch.uint8
    B, H, W = (x.shape[0], x.shape[1], x.shape[2]) if u8 else (x.shape[0], x.shape[2], x.shape[3])
    w5 = pack_head5(w, torch.float16)
    scale = torch.ones(16, dtype=torch.float32, device=x.device)
    shift = torch.zeros(16, dtype=torch.float32, device=x.device)
    shift[:bias.numel()] = bias.float()
    out = torch.empty(B, H, W, 16, dtype=torch.float16, device=x.device)
    check(_lib.load().cfen_head_conv5(1, int(u8), ptr(x), ptr(w5), ptr(scale), ptr(shift), ptr(out), B, H, W, 16, act, current_stream()), "head_conv5")
    return out


def gemm_chain(phases, M, team=48, fold=None, sync=None):
    """cfen_gemm_chain: a list of dependent GEMM phases in one persistent launch.  Every phase is a dict with x [M,K], w [N,K] (row-major; packed
    to a fragment stream here) or w_stream, y (preallocated output, [M,N] or the NHWC map when fold), and optionally bias, lnf_s, residual, pos,
    relu, nsplit, fold.  fold = (H, W, cs, C, p) of the map.  Returns the error word (0 = fine)."""
    from .packing import pack_stream_tiles
    from ._lib import ChainArgsC
    a = ChainArgsC()
    keep = []
    a.nphases, a.M = len(phases), M
    maxslab = 0
    for i, ph in enumerate(phases):
        x, y = ph["x"], ph["y"]
        ws = ph.get("w_stream")
        if ws is None:
            ws = pack_stream_tiles(ph["w"].contiguous())
        keep.append(ws)
        N, K = (ph["w"].shape if "w" in ph else (ph["N"], ph["K"]))
        _cuda(x, ws, y, ph.get("bias"), ph.get("lnf_s"), ph.get("residual"), ph.get("pos"))
        q = a.phase[i]
        q.x, q.w_stream, q.bias, q.lnf_s, q.residual, q.pos, q.y = (ptr(x).value, ptr(ws).value, ptr(ph.get("bias")).value, ptr(ph.get("lnf_s")).value,
                                                                  ptr(ph.get("residual")).value, ptr(ph.get("pos")).value, ptr(y).value)
        q.ldx, q.ldr, q.ldy = x.shape[1], N, N
        q.period = ph["pos"].shape[0] if ph.get("pos") is not None else 0
        q.N, q.K, q.relu, q.nsplit, q.fold = N, K, int(ph.get("relu", False)), int(ph.get("nsplit", 1)), int(ph.get("fold", False))
        maxslab = max(maxslab, (N // 128) * q.nsplit)
    if fold is not None:
        a.fold_H, a.fold_W, a.fold_cs, a.fold_C, a.fold_p = fold
    own = sync is None
    if own:
        sync = torch.zeros(8192 + ((M + 127) // 128) * maxslab * 65536, dtype=torch.uint8, device=phases[0]["x"].device)
    a.sync_ws, a.sync_ws_bytes = ptr(sync).value, sync.numel()
    check(_lib.load().cfen_gemm_chain(dtype_code(phases[0]["x"].dtype), ctypes.byref(a), team, current_stream()), "gemm_chain")
    return int(sync[:8].view(torch.int32)[1].item()) if own else None     # with a caller-owned `sync` buffer: no read-back (word 1 of it = error)


def layernorm(x, gamma, beta, eps=1e-5):
    _cuda(x, gamma, beta)
    out = torch.empty_like(x)
    check(_lib.load().cfen_layernorm(dtype_code(x.dtype), ptr(x), ptr(out), ptr(gamma), ptr(beta), x.shape[0], x.shape[1], eps,
                                     current_stream()), "layernorm")
    return out


def attention(qkv, nseq, S, heads):
    """qkv: [nseq*S, 3*D] -> [nseq*S, D]"""
    _cuda(qkv)
    D = qkv.shape[1] // 3
    out = torch.empty(qkv.shape[0], D, dtype=qkv.dtype, device=qkv.device)
    check(_lib.load().cfen_attention(dtype_code(qkv.dtype), ptr(qkv), ptr(out), nseq, S, heads, D // heads, current_stream()), "attention")
    return out


def attention_head_major(qkv, nseq, S, heads):
    """qkv in the head-major layout of embed_qkv(head_major_heads=heads) -> [nseq*S, D] row-major"""
    _cuda(qkv)
    D = qkv.numel() // (3 * nseq * S)
    out = torch.empty(nseq * S, D, dtype=qkv.dtype, device=qkv.device)
    check(_lib.load().cfen_attention_head_major(dtype_code(qkv.dtype), ptr(qkv), ptr(out), nseq, S, heads, D // heads, current_stream()),
          "attention_head_major")
    return out


def mlp_block(x, w1a, b1a, w2a, b2a, ln=None, second=None, fold=None, proj=None):
    """Fused y1 = x + W2a relu(W1a LN(x)+b1a) + b2a [; y2 = y1 + W2b relu(W1b y1 + b1b) + b2b].
    Weights must already carry packing.kperm32 on their k axis for fp16.  fold = (B, H, W, C, cs, ws, p) writes
    the result into a fresh NHWC map instead of a token matrix.  proj = (att, w_proj): x is first replaced by x + att @ w_proj.T."""
    _cuda(x, w1a, b1a, w2a, b2a)
    M, D = x.shape
    a = MlpArgsC(x=x.data_ptr(), w1a=w1a.data_ptr(), b1a=b1a.data_ptr(), w2a=w2a.data_ptr(), b2a=b2a.data_ptr(), M=M, D=D,
                 H=w1a.shape[0], eps=1e-5)
    if proj is not None:
        _cuda(*proj)
        a.att, a.w_proj = proj[0].data_ptr(), proj[1].data_ptr()
    if ln is not None:
        _cuda(*ln)
        a.ln_gamma, a.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
    if second is not None:
        _cuda(*second)
        a.w1b, a.b1b, a.w2b, a.b2b = (t.data_ptr() for t in second)
    if fold is None:
        out = torch.empty_like(x)
        a.y = out.data_ptr()
    else:
        B, H, W, C, cs, ws, p = fold
        out = torch.zeros(B, H, W, cs, dtype=x.dtype, device=x.device)
        a.fmap, a.mapH, a.mapW, a.C, a.cs, a.ws, a.p = out.data_ptr(), H, W, C, cs, ws, p
    check(_lib.load().cfen_mlp_block(dtype_code(x.dtype), ctypes.byref(a), current_stream()), "mlp_block")
    return out


def mlp_stream_block(x, wa, b1a, b2a, hidden, ln=None, second=None, fold=None, proj=None):
    """cfen_mlp_stream_block (csrc/k_stream.hip): mlp_block with the matrices as fragment streams.  wa = packing.pack_stream_pair(W1k, W2k);
    second = (wb, b1b, b2b); proj = (att, packing.pack_stream_sq(Wp)); fold = (B, H, W, C, cs, ws, p)."""
    from ._lib import MlpStreamArgsC
    _cuda(x, wa, b1a, b2a)
    M, D = x.shape
    a = MlpStreamArgsC(x=x.data_ptr(), wa_stream=wa.data_ptr(), b1a=b1a.data_ptr(), b2a=b2a.data_ptr(), M=M, D=D, H=hidden, eps=1e-5)
    if proj is not None:
        _cuda(*proj)
        a.att, a.wp_stream = proj[0].data_ptr(), proj[1].data_ptr()
    if ln is not None:
        _cuda(*ln)
        a.ln_gamma, a.ln_beta = ln[0].data_ptr(), ln[1].data_ptr()
    if second is not None:
        _cuda(*second)
        a.wb_stream, a.b1b, a.b2b = (t.data_ptr() for t in second)
    if fold is None:
        out = torch.empty_like(x)
        a.y = out.data_ptr()
    else:
        B, H, W, C, cs, ws, p = fold
        out = torch.zeros(B, H, W, cs, dtype=x.dtype, device=x.device)
        a.fmap, a.mapH, a.mapW, a.C, a.cs, a.ws, a.p = out.data_ptr(), H, W, C, cs, ws, p
    check(_lib.load().cfen_mlp_stream_block(dtype_code(x.dtype), ctypes.byref(a), current_stream()), "mlp_stream_block")
    return out


def lvit_window(fmap, C, ws, p, packed, name, hidden, cs_out=None, eps=1e-5):
    """whole LViT block (C = 24, p = 2, ws = 32) map -> map in one launch; `packed` = packing.pack_vit entries of `name` + packing.pack_lvit_window's fragment stream (`hidden` = the stream's hidden width)"""
    from ._lib import LvitArgsC
    _cuda(fmap)
    B, H, W, cs = fmap.shape
    cs_out = cs if cs_out is None else cs_out
    out = torch.zeros(B, H, W, cs_out, dtype=fmap.dtype, device=fmap.device)
    g = lambda k: packed[name + k].data_ptr()
    a = LvitArgsC(fmap=fmap.data_ptr(), out=out.data_ptr(), B=B, H=H, W=W, C=C, cs_in=cs, cs_out=cs_out, ws=ws, p=p,
                  w_stream=g(".lw.ws"), be=g(".embed.b"), pos=g(".pos"), ln1_gamma=g(".ln1.g"), ln1_beta=g(".ln1.b"),
                  ln2_gamma=g(".ln2.g"), ln2_beta=g(".ln2.b"), b1a=g(".ffn1.b"), b2a=g(".ffn2.b"), b1b=g(".head1.b"), b2b=g(".head2.b"),
                  hidden=hidden, eps=eps)
    check(_lib.load().cfen_lvit_window(dtype_code(fmap.dtype), ctypes.byref(a), current_stream()), "lvit_window")
    return out


def patchify(fmap, C, ws, p, pool=1):
    """fmap: NHWC [B,Hf,Wf,cs] -> tokens [B*nwin*S, p*p*C] in (i,j,c) feature order."""
    _cuda(fmap)
    B, Hf, Wf, cs = fmap.shape
    H, W = Hf // pool, Wf // pool
    tok = torch.empty(B * H * W // (p * p), p * p * C, dtype=fmap.dtype, device=fmap.device)
    check(_lib.load().cfen_patchify(dtype_code(fmap.dtype), ptr(fmap), ptr(tok), B, H, W, C, cs, ws, p, pool, current_stream()), "patchify")
    return tok


def embed_gather(fmap, C, ws, p, w, bias, pos):
    """tokens = patchify(fmap) gathered inside the GEMM:  tok @ w.T + bias + tok + pos[row % len(pos)]  -> [M, p*p*C]"""
    _cuda(fmap, w, bias, pos)
    B, H, W, cs = fmap.shape
    D = p * p * C
    out = torch.empty(B * H * W // (p * p), D, dtype=fmap.dtype, device=fmap.device)
    check(_lib.load().cfen_embed_gather(dtype_code(fmap.dtype), ptr(fmap), B, H, W, C, cs, ws, p, ptr(w), w.shape[1], ptr(bias), ptr(pos),
                                        pos.shape[0] if pos is not None else 0, ptr(out), D, current_stream()), "embed_gather")
    return out


def embed_qkv(fmap, C, ws, p, we, be, pos, ln_g, ln_b, wqkv, eps=1e-5, head_major_heads=0, stream_weights=False):
    """fused LViT front half (D = p*p*C in {96,192}); we / wqkv with the k axis in packing.kperm32 order for fp16.
    Returns (x1 [M,D], qkv [M,3D])."""
    from ._lib import EmbedQkvArgsC
    _cuda(fmap, we, be, pos, ln_g, ln_b, wqkv)
    B, H, W, cs = fmap.shape
    D = p * p * C
    M = B * H * W // (p * p)
    x1 = torch.empty(M, D, dtype=fmap.dtype, device=fmap.device)
    qkv = torch.empty(M, 3 * D, dtype=fmap.dtype, device=fmap.device)
    a = EmbedQkvArgsC(fmap=fmap.data_ptr(), B=B, H=H, W=W, C=C, cs=cs, ws=ws, p=p, we=we.data_ptr(), be=be.data_ptr(), pos=pos.data_ptr(),
                      ln_gamma=ln_g.data_ptr(), ln_beta=ln_b.data_ptr(), wqkv=wqkv.data_ptr(), x1=x1.data_ptr(), qkv=qkv.data_ptr(), eps=eps,
                      head_major_heads=head_major_heads)
    fn = _lib.load().cfen_embed_qkv_stream if stream_weights else _lib.load().cfen_embed_qkv   # stream_weights: we / wqkv = packing.pack_stream_rows (D = 384)
    check(fn(dtype_code(fmap.dtype), ctypes.byref(a), current_stream()), "embed_qkv")
    return x1, qkv


def unpatchify(tok, B, H, W, C, cs, ws, p):
    _cuda(tok)
    fmap = torch.zeros(B, H, W, cs, dtype=tok.dtype, device=tok.device)
    check(_lib.load().cfen_unpatchify(dtype_code(tok.dtype), ptr(tok), ptr(fmap), B, H, W, C, cs, ws, p, current_stream()), "unpatchify")
    return fmap


def upsample4(small, cs_out=None):
    _cuda(small)
    B, h, w, cs_in = small.shape
    cs_out = cs_out or cs_in
    out = torch.zeros(B, 4 * h, 4 * w, cs_out, dtype=small.dtype, device=small.device)
    check(_lib.load().cfen_upsample4(dtype_code(small.dtype), ptr(small), ptr(out), B, h, w, cs_in, cs_in, cs_out, current_stream()), "upsample4")
    return out


def nchw_to_nhwc(x, cs, dtype):
    _cuda(x)
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, cs, dtype=dtype, device=x.device)
    check(_lib.load().cfen_nchw_to_nhwc(dtype_code(dtype), ptr(x), ptr(out), B, C, H, W, cs, current_stream()), "nchw_to_nhwc")
    return out


def conv2d(src0, weight, scale, shift, cin, cout, k=3, stride=1, pad=1, reflect=False, src1=None, transpose=False, act=0,
           res0=None, res1=None, cs_out=None, nchw_f32=False, rows_layout=False, toeplitz=False, src2=None):
    """src*: NHWC [B,H,W,cs]; weight/scale/shift packed by packing.pack_conv*_weight / affine.
    rows_layout: weight from packing.pack_conv_weight_rows -> the LDS-tiled stride-1 kernel."""
    _cuda(src0, src1, src2, weight, scale, shift, res0, res1)
    B, Hin, Win, cs_in = src0.shape
    cout_pad = round_up(cout, 16)
    if transpose:
        Hout, Wout = 2 * Hin, 2 * Win
    else:
        Hout, Wout = (Hin + 2 * pad - k) // stride + 1, (Win + 2 * pad - k) // stride + 1
    if nchw_f32:
        out = torch.empty(B, cout, Hout, Wout, dtype=torch.float32, device=src0.device)
        cs_out = cout_pad
    else:
        cs_out = cs_out or round_up(cout, 8)
        out = torch.zeros(B, Hout, Wout, cs_out, dtype=src0.dtype, device=src0.device)
    a = ConvArgsC(kind=1 if transpose else 0, k=k, stride=stride, pad=pad, reflect=int(reflect), nsrc=3 if src2 is not None else 2 if src1 is not None else 1,
                  B=B, Hin=Hin, Win=Win, Cin=cin, cs_in=cs_in, Cout=cout, Cout_pad=cout_pad, Kpad=weight.shape[-1], cs_out=cs_out,
                  act=act, out_nchw_f32=int(nchw_f32), cs_res=cs_out, wlayout=2 if toeplitz else int(rows_layout),
                  src0=src0.data_ptr(), src1=src1.data_ptr() if src1 is not None else None, weight=weight.data_ptr(),
                  scale=scale.data_ptr(), shift=shift.data_ptr(), res0=res0.data_ptr() if res0 is not None else None,
                  res1=res1.data_ptr() if res1 is not None else None, out=out.data_ptr(),
                  src2=src2.data_ptr() if src2 is not None else None)
    check(_lib.load().cfen_conv2d(dtype_code(src0.dtype), ctypes.byref(a), current_stream()), "conv2d")
    return out


def _stats_ws(B, device):
    n = _lib.load().cfen_stats_workspace(B, 128)
    return torch.empty(n // 4, dtype=torch.float32, device=device)


def instnorm_relu_(x, C, eps=1e-5):
    """in place on NHWC [B,H,W,cs]"""
    _cuda(x)
    B, H, W, cs = x.shape
    ws = _stats_ws(B, x.device)
    check(_lib.load().cfen_instnorm_relu(dtype_code(x.dtype), ptr(x), ptr(ws), B, H * W, C, cs, eps, current_stream()), "instnorm_relu")
    return x


def cfsm2g(x0, x1, x2, w, C):
    _cuda(x0, x1, x2, w)
    B, H, W, cs = x0.shape
    out = torch.zeros_like(x0)
    ws = _stats_ws(B, x0.device)
    check(_lib.load().cfen_cfsm2g(dtype_code(x0.dtype), ptr(x0), ptr(x1), ptr(x2), ptr(out), ptr(w), ptr(ws), B, H * W, C, cs,
                                  current_stream()), "cfsm2g")
    return out


def tensor2im_u8(x):
    """(1|3,H,W) fp32 CUDA tensor in [-1,1] -> (H,W,3) uint8 CUDA tensor; util.tensor2im on the device"""
    _cuda(x)
    if x.dim() != 3 or x.dtype != torch.float32 or not x.is_contiguous():
        raise ValueError("tensor2im_u8 needs a contiguous (C,H,W) float32 tensor")
    C, H, W = x.shape
    out = torch.empty(H, W, 3, dtype=torch.uint8, device=x.device)
    check(_lib.load().cfen_tensor2im_u8(ptr(x), ptr(out), C, H, W, current_stream()), "tensor2im_u8")
    return out


def u8hwc_to_nhwc(img, cs, dtype):
    """(B,H,W,3) uint8 CUDA tensor -> normalised NHWC [B,H,W,cs] of `dtype` (ToTensor + Normalize(0.5, 0.5) + layout)"""
    _cuda(img)
    if img.dim() != 4 or img.shape[3] != 3 or img.dtype != torch.uint8 or not img.is_contiguous():
        raise ValueError("u8hwc_to_nhwc needs a contiguous (B,H,W,3) uint8 tensor")
    B, H, W, _ = img.shape
    out = torch.empty(B, H, W, cs, dtype=dtype, device=img.device)
    check(_lib.load().cfen_u8hwc_to_nhwc(dtype_code(dtype), ptr(img), ptr(out), B, H, W, cs, current_stream()), "u8hwc_to_nhwc")
    return out


def to_nhwc(x, cs=None, dtype=None):
    """NCHW torch tensor -> zero-padded NHWC (test helper; plain torch, not on the product path)."""
    B, C, H, W = x.shape
    cs = cs or round_up(C, 8)
    out = torch.zeros(B, H, W, cs, dtype=dtype or x.dtype, device=x.device)
    out[..., :C] = x.permute(0, 2, 3, 1)
    return out


def from_nhwc(x, C):
    return x[..., :C].permute(0, 3, 1, 2).contiguous()
