"""GPU parity of the HIP deformable-conv operator (through the reference-shaped Python API and the C ABI)
against the C oracle, on the feature-map shapes of the v3 generator and the edge cases
the reference's Python layer guards."""
import pytest
import torch
import torch.nn.functional as F

import dcn_oracle
from cfen_vit_dehazing_amd import dcn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 2e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("B,C,H,Cout,k,stride,pad,dil,groups,dg", [
    (2, 24, 32, 24, 3, 1, 1, 1, 1, 1), (1, 48, 16, 48, 3, 1, 1, 1, 1, 8), (1, 96, 16, 96, 3, 1, 1, 1, 1, 8),
    (2, 8, 13, 6, 3, 2, 1, 1, 1, 2), (1, 8, 12, 12, 5, 1, 2, 1, 2, 1), (1, 6, 10, 160, 3, 1, 2, 2, 1, 3), (3, 3, 9, 5, 1, 1, 0, 1, 1, 1),
    # k_dcn_lean: two conv groups of 24 channels cut into deformable groups of 12, stride / dilation 2 with 40 output rows and groups of 3,
    # two output-row blocks, a 5x5 kernel (three tap slices, the last one padded), a unit inside a 48-channel deformable group
    (1, 48, 16, 48, 3, 1, 1, 1, 2, 4), (2, 24, 17, 40, 3, 2, 2, 2, 1, 8), (1, 24, 12, 160, 3, 1, 1, 1, 1, 1), (1, 24, 14, 24, 5, 1, 2, 1, 1, 4),
    (1, 96, 12, 48, 3, 1, 1, 1, 1, 2)])
def test_deform_conv_v1_and_v2(dtype, B, C, H, Cout, k, stride, pad, dil, groups, dg):
    x, w = rnd((B, C, H, H + 3), 1), rnd((Cout, C // groups, k, k), 2, (C // groups * k * k) ** -0.5)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (H + 3 + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = rnd((B, dg * 2 * k * k, Ho, Wo), 3, 2.0)
    mask = torch.sigmoid(rnd((B, dg * k * k, Ho, Wo), 4))
    bias = rnd((Cout,), 5)
    xq, wq, oq, mq, bq = (t.to(dtype) for t in (x, w, off, mask, bias))
    want1 = dcn_oracle.deform_conv(xq.float(), oq.float(), wq.float(), stride, pad, dil, groups, dg)
    got1 = dcn.deform_conv(xq.to(DEV), oq.to(DEV), wq.to(DEV), stride, pad, dil, groups, dg)
    assert got1.dtype == dtype and got1.shape == want1.shape
    assert float((got1.float().cpu() - want1).abs().max()) <= tol(dtype)
    want2 = dcn_oracle.deform_conv(xq.float(), oq.float(), wq.float(), stride, pad, dil, groups, dg, mask=mq.float(), bias=bq.float())
    got2 = dcn.modulated_deform_conv(xq.to(DEV), oq.to(DEV), mq.to(DEV), wq.to(DEV), bq.to(DEV), stride, pad, dil, groups, dg)
    assert float((got2.float().cpu() - want2).abs().max()) <= tol(dtype)


@pytest.mark.parametrize("C,dg", [(24, 1), (24, 8), (48, 8), (96, 8)])
def test_lean_kernel_and_round2_kernel_agree(C, dg):
    """`dcn.tile` 0 runs k_dcn_nhwc (round 2) on the shapes k_dcn_lean serves by default: same sampling rules; the lean kernel interpolates
    fp16 maps with packed fp16 FMAs (as the reference's half instantiation does), k_dcn_nhwc in fp32 -- equal to fp16 rounding of the samples."""
    from cfen_vit_dehazing_amd import ops
    H = 40
    x, w = rnd((2, C, H, H), 21).half().to(DEV), rnd((C, C, 3, 3), 22, (C * 9) ** -0.5).half().to(DEV)
    off = rnd((2, dg * 18, H, H), 23, 3.0).half().to(DEV)
    mask = torch.sigmoid(rnd((2, dg * 9, H, H), 24)).half().to(DEV)
    bias = rnd((C,), 25).half().to(DEV)
    try:
        got = [(dcn.deform_conv(x, off, w, 1, 1, 1, 1, dg), dcn.modulated_deform_conv(x, off, mask, w, bias, 1, 1, 1, 1, dg))
               for t in (1, 0) if ops.tune("dcn.tile", t) is None]
    finally:
        ops.tune("dcn.tile", 1)
    for a, b in zip(got[0], got[1]):
        assert float((a.float() - b.float()).abs().max()) <= 8e-3


@pytest.mark.parametrize("dtype,C,dg", [(torch.float16, 24, 1), (torch.float16, 48, 8), (torch.float32, 24, 4)])
def test_channels_last_input_skips_the_layout_pass_and_gives_the_same_bits(dtype, C, dg):
    """(extension, round 6) an undifferentiated call on a channels_last input samples from the tensor's own NHWC memory (cfen_*_forward_nhwc): bit for bit the result of
    the contiguous call, v1 and v2; a differentiated call keeps the NCHW copy (its backward kernels read NCHW)"""
    d = "cuda:0"
    g = torch.Generator().manual_seed(7)
    B, H = 2, 40
    x = torch.randn(B, C, H, H, generator=g).to(dtype).to(d)
    w = (torch.randn(C, C, 3, 3, generator=g) * (C * 9) ** -0.5).to(dtype).to(d)
    off = (torch.randn(B, dg * 18, H, H, generator=g) * 2.0).to(dtype).to(d)
    mask = torch.sigmoid(torch.randn(B, dg * 9, H, H, generator=g)).to(dtype).to(d)
    bias = torch.randn(C, generator=g).to(dtype).to(d)
    xcl = x.contiguous(memory_format=torch.channels_last)
    assert not xcl.is_contiguous()
    with torch.no_grad():
        assert torch.equal(dcn.deform_conv(xcl, off, w, 1, 1, 1, 1, dg), dcn.deform_conv(x, off, w, 1, 1, 1, 1, dg))
        assert torch.equal(dcn.modulated_deform_conv(xcl, off, mask, w, bias, 1, 1, 1, 1, dg), dcn.modulated_deform_conv(x, off, mask, w, bias, 1, 1, 1, 1, dg))
    xg = xcl.detach().clone(memory_format=torch.preserve_format).requires_grad_()
    y = dcn.deform_conv(xg, off, w, 1, 1, 1, 1, dg)
    y.float().sum().backward()
    assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_pack_modules_at_init_are_plain_convs():
    # DeformConvPack zero-initialises conv_offset (deform_conv.py:211-213) => plain conv
    torch.manual_seed(0)
    m = dcn.DeformConvPack(24, 24, 3, stride=1, padding=1, deformable_groups=8).to(DEV)
    x = rnd((2, 24, 32, 32), 1).to(DEV)
    want = F.conv2d(x.cpu(), m.weight.detach().cpu(), padding=1)
    assert float((m(x).cpu() - want).abs().max()) <= 2e-4
    m2 = dcn.ModulatedDeformConvPack(24, 12, 3, stride=1, padding=1, deformable_groups=2, bias=True).to(DEV)
    with torch.no_grad():
        m2.bias.copy_(rnd((12,), 2))
    want2 = 0.5 * F.conv2d(x.cpu(), m2.weight.detach().cpu(), padding=1) + m2.bias.detach().cpu().view(1, -1, 1, 1)   # sigmoid(0) = 0.5
    assert float((m2(x).cpu() - want2).abs().max()) <= 2e-4
    m3 = dcn.ModulatedDeformConvPack2(24, 12, 3, stride=1, padding=1, extra_offset_mask=True, offset_in_channel=8).to(DEV)
    feat = rnd((2, 8, 32, 32), 3).to(DEV)
    assert m3([x, feat]).shape == (2, 12, 32, 32)


def test_error_behaviour_matches_reference():
    x = torch.zeros(2, 4, 8, 8, device=DEV)
    w = torch.zeros(4, 4, 3, 3, device=DEV)
    with pytest.raises(ValueError):
        dcn.deform_conv(torch.zeros(4, 8, 8, device=DEV), torch.zeros(1, device=DEV), w)            # not 4-D (deform_conv.py:19-21)
    with pytest.raises(ValueError):
        dcn.deform_conv(torch.zeros(1, 4, 2, 2, device=DEV), torch.zeros(1, 18, 1, 1, device=DEV), w)  # output too small (:91-93)
    with pytest.raises(AssertionError):
        dcn.deform_conv(torch.zeros(3, 4, 8, 8, device=DEV), torch.zeros(3, 18, 8, 8, device=DEV), w, 1, 1, 1, 1, 1, 2)  # step !| batch (:40)
    with pytest.raises(NotImplementedError):
        dcn.deform_conv(x.cpu(), torch.zeros(2, 18, 8, 8), w.cpu(), 1, 1)                           # CPU tensors (:36-37)
    with pytest.raises(RuntimeError):
        dcn.deform_conv(x, torch.zeros(2, 16, 8, 8, device=DEV), w, 1, 1)                            # offset channels (.cpp:129-130)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("C,H,dg", [(24, 256, 1), (24, 256, 8), (48, 128, 1), (48, 128, 8), (96, 64, 1), (96, 64, 8)])
def test_deform_conv_at_the_generator_feature_map_shapes(dtype, C, H, dg):
    """(B,24,256,256) / (B,48,128,128) / (B,96,64,64), 3x3 s1 p1, deformable groups 1 and 8, v1 and v2.
    Batch 2 (the oracle is scalar C); offsets of +-2 pixels reach over the borders."""
    B = 2
    x, w = rnd((B, C, H, H), 11), rnd((C, C, 3, 3), 12, (C * 9) ** -0.5)
    off = rnd((B, dg * 18, H, H), 13, 2.0)
    mask = torch.sigmoid(rnd((B, dg * 9, H, H), 14))
    bias = rnd((C,), 15)
    xq, wq, oq, mq, bq = (t.to(dtype) for t in (x, w, off, mask, bias))
    want1 = dcn_oracle.deform_conv(xq.float(), oq.float(), wq.float(), 1, 1, 1, 1, dg)
    got1 = dcn.deform_conv(xq.to(DEV), oq.to(DEV), wq.to(DEV), 1, 1, 1, 1, dg)
    assert float((got1.float().cpu() - want1).abs().max()) <= tol(dtype)
    want2 = dcn_oracle.deform_conv(xq.float(), oq.float(), wq.float(), 1, 1, 1, 1, dg, mask=mq.float(), bias=bq.float())
    got2 = dcn.modulated_deform_conv(xq.to(DEV), oq.to(DEV), mq.to(DEV), wq.to(DEV), bq.to(DEV), 1, 1, 1, 1, dg)
    assert float((got2.float().cpu() - want2).abs().max()) <= tol(dtype)


# ---- backward (csrc/k_dcn_bwd.hip) through autograd of the reference-shaped functions, against oracle/dcn_oracle.c -----------------------

def _grad_close(got, want, dtype, what):
    scale = max(1.0, float(want.abs().max()))
    d = float((got.float().cpu() - want).abs().max())
    bar = (3e-4 if dtype == torch.float32 else 2e-2) * scale
    assert d <= bar, "%s: max-abs %.3e > %.1e (scale %.2f)" % (what, d, bar, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("B,C,H,Cout,k,stride,pad,dil,groups,dg", [
    (2, 24, 16, 24, 3, 1, 1, 1, 1, 1), (1, 48, 12, 48, 3, 1, 1, 1, 1, 8), (2, 8, 13, 6, 3, 2, 1, 1, 1, 2), (1, 8, 12, 12, 5, 1, 2, 1, 2, 1),
    (1, 6, 10, 20, 3, 1, 2, 2, 1, 3), (3, 3, 9, 5, 1, 1, 0, 1, 1, 1)])
def test_deform_conv_backward_v1_and_v2(dtype, B, C, H, Cout, k, stride, pad, dil, groups, dg):
    x, w = rnd((B, C, H, H + 3), 1), rnd((Cout, C // groups, k, k), 2, (C // groups * k * k) ** -0.5)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (H + 3 + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off, mask, bias = rnd((B, dg * 2 * k * k, Ho, Wo), 3, 1.5), torch.rand(B, dg * k * k, Ho, Wo, generator=torch.Generator().manual_seed(4)), rnd((Cout,), 5)
    gy = rnd((B, Cout, Ho, Wo), 6)
    if dtype == torch.float16:                  # the oracle sees what the kernel sees
        x, w, off, mask, gy = (t.half().float() for t in (x, w, off, mask, gy))
    leaf = lambda t: t.to(DEV).to(dtype).requires_grad_()
    # DCNv1
    xi, oi, wi = leaf(x), leaf(off), leaf(w)
    y = dcn.deform_conv(xi, oi, wi, stride, pad, dil, groups, dg)
    y.backward(gy.to(DEV).to(dtype))
    want = dcn_oracle.deform_conv_backward(x, off, w, gy, stride, pad, dil, groups, dg)
    _grad_close(xi.grad, want["input"], dtype, "v1 grad_input")
    _grad_close(oi.grad, want["offset"], dtype, "v1 grad_offset")
    _grad_close(wi.grad, want["weight"], dtype, "v1 grad_weight")
    # DCNv2 (+ mask, + bias)
    xi, oi, mi, wi, bi = leaf(x), leaf(off), leaf(mask), leaf(w), leaf(bias)
    y = dcn.modulated_deform_conv(xi, oi, mi, wi, bi, stride, pad, dil, groups, dg)
    y.backward(gy.to(DEV).to(dtype))
    want = dcn_oracle.deform_conv_backward(x, off, w, gy, stride, pad, dil, groups, dg, mask=mask, with_bias=True)
    for name, t in (("input", xi), ("offset", oi), ("mask", mi), ("weight", wi), ("bias", bi)):
        _grad_close(t.grad, want[name], dtype, "v2 grad_" + name)


def test_deform_conv_backward_only_weight_or_only_input_is_requested():
    """dcn/deform_conv.py:62-80: the two extension calls are made independently, by needs_input_grad"""
    x, w, off = rnd((1, 8, 10, 10), 1), rnd((8, 8, 3, 3), 2, 0.2), rnd((1, 18, 10, 10), 3)
    gy = rnd((1, 8, 10, 10), 4)
    want = dcn_oracle.deform_conv_backward(x, off, w, gy, 1, 1, 1, 1, 1)
    wi = w.to(DEV).requires_grad_()
    dcn.deform_conv(x.to(DEV), off.to(DEV), wi, 1, 1, 1, 1, 1).backward(gy.to(DEV))
    _grad_close(wi.grad, want["weight"], torch.float32, "grad_weight alone")
    xi = x.to(DEV).requires_grad_()
    dcn.deform_conv(xi, off.to(DEV), w.to(DEV), 1, 1, 1, 1, 1).backward(gy.to(DEV))
    _grad_close(xi.grad, want["input"], torch.float32, "grad_input alone")


def test_deform_conv_pack_trains_one_sgd_step_like_conv2d_at_init():
    """DeformConvPack at init (zero offset conv) is a plain conv: its weight gradient equals conv2d's (deform_conv.py:222-231)"""
    torch.manual_seed(0)
    m = dcn.DeformConvPack(8, 8, 3, stride=1, padding=1, deformable_groups=2).to(DEV)
    x = rnd((2, 8, 12, 12), 7).to(DEV)
    y = m(x)
    y.square().mean().backward()
    w = m.weight.detach().clone().requires_grad_()
    F.conv2d(x, w, None, 1, 1).square().mean().backward()
    assert float((m.weight.grad - w.grad).abs().max()) <= 2e-5


@pytest.mark.parametrize("off_scale", [1.0, 12.0])
def test_deform_conv_backward_lds_col2im_equals_global_atomics(off_scale):
    """grad_input through the LDS-privatised col2im with parked tiles (1, default), with tiles flushed by global atomics (2) and through
    plain global atomics (0); small offsets (all inside the tile halo) and offsets far beyond it (the per-add fallback to global memory),
    40 x 36 pixels so that tiles are ragged; all against the oracle"""
    from cfen_vit_dehazing_amd import _lib
    x, w = rnd((2, 12, 40, 36), 1), rnd((12, 12, 3, 3), 2, 0.15)
    off, gy = rnd((2, 36, 40, 36), 3, off_scale), rnd((2, 12, 40, 36), 4)
    want = dcn_oracle.deform_conv_backward(x, off, w, gy, 1, 1, 1, 1, 2)
    lib = _lib.load()
    got = {}
    for lds in (1, 2, 0):
        old = lib.cfen_deform_conv_backward_set_lds(lds)
        try:
            xi = x.to(DEV).requires_grad_()
            dcn.deform_conv(xi, off.to(DEV), w.to(DEV), 1, 1, 1, 1, 2).backward(gy.to(DEV))
            got[lds] = xi.grad.cpu()
        finally:
            lib.cfen_deform_conv_backward_set_lds(old)
        _grad_close(got[lds], want["input"], torch.float32, "grad_input (lds=%d)" % lds)
    for lds in (1, 2):
        assert float((got[lds] - got[0]).abs().max()) <= 1e-4 * max(1.0, float(want["input"].abs().max()))


def test_deform_conv_backward_fixed_point_col2im_edge_cases():
    """k_dcnb_col2im_lds accumulates in 64-bit fixed point scaled by the workgroup's largest contribution: the result must not depend on
    the magnitude of the gradients (power-of-two scalings are exact, so grad_input scales BITWISE), an all-zero grad_output gives exact zeros,
    run-to-run results are bit-identical (the adds are order-independent; offsets stay inside the tile halo so no global atomics are
    involved), and a non-finite grad_output is not laundered into finite numbers."""
    x, w = rnd((2, 16, 24, 20), 1), rnd((8, 16, 3, 3), 2, 0.1)
    off, gy = rnd((2, 18, 24, 20), 3, 1.0), rnd((2, 8, 24, 20), 4)

    def grad_in(g):
        xi = x.to(DEV).requires_grad_()
        dcn.deform_conv(xi, off.to(DEV), w.to(DEV), 1, 1, 1, 1, 1).backward(g.to(DEV))
        return xi.grad.cpu()

    base = grad_in(gy)
    _grad_close(base, dcn_oracle.deform_conv_backward(x, off, w, gy, 1, 1, 1, 1, 1)["input"], torch.float32, "grad_input")
    assert torch.equal(grad_in(gy), base)                                        # deterministic
    for k in (-60, -20, 20, 60):
        assert torch.equal(grad_in(gy * 2.0 ** k), base * 2.0 ** k), "scale 2^%d" % k
    assert torch.equal(grad_in(torch.zeros_like(gy)), torch.zeros_like(base))
    for poison in (float("inf"), float("nan")):
        bad = gy.clone()
        bad[1, 3, 10, 7] = poison
        g = grad_in(bad)
        assert not bool(torch.isfinite(g[1]).all()) and bool(torch.isfinite(g[0]).all())
