"""Pin the C oracle of the deformable conv (oracle/dcn_oracle.c) with the known-answer properties that
follow from the reference kernel code: the reference ships no DCN tests or vectors and its
CUDA extension cannot be built here, so these are the only pins (stated in the oracle's header)."""
import math

import pytest
import torch
import torch.nn.functional as F

import dcn_oracle


def rnd(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


@pytest.mark.parametrize("k,stride,pad,dil,groups", [(3, 1, 1, 1, 1), (3, 2, 1, 1, 1), (3, 1, 2, 2, 1), (5, 1, 2, 1, 2), (1, 1, 0, 1, 1)])
def test_zero_offsets_equal_conv2d(k, stride, pad, dil, groups):
    # integer sample points give bilinear weights (1,0,0,0); the (-1,H) bound reproduces zero padding (.cu:226-236)
    x, w = rnd((2, 8, 11, 13), 1), rnd((6, 8 // groups, k, k), 2, 0.2)
    want = F.conv2d(x, w, None, stride, pad, dil, groups)
    off = torch.zeros(2, 2 * k * k, want.shape[2], want.shape[3])
    got = dcn_oracle.deform_conv(x, off, w, stride, pad, dil, groups, 1)
    assert torch.allclose(got, want, atol=1e-5)


def test_integer_offsets_equal_shifted_taps():
    x, w = rnd((1, 4, 12, 12), 3), rnd((5, 4, 3, 3), 4, 0.3)
    off = torch.zeros(1, 18, 12, 12)
    off[:, 0::2] = 2.0          # every tap samples 2 rows lower
    off[:, 1::2] = -1.0         # and 1 column to the left
    got = dcn_oracle.deform_conv(x, off, w, 1, 1, 1, 1, 1)
    xs = torch.zeros(1, 4, 12 + 8, 12 + 8)
    xs[:, :, 4:16, 4:16] = x
    # shifted image with zeros outside: out(y,x) = conv(x_shift), x_shift(y,x) = x(y+2, x-1)
    x_shift = xs[:, :, 6:18, 3:15]
    want = F.conv2d(x_shift, w, padding=1)
    # taps that land outside the ORIGINAL image are zero in both; but conv's own zero padding ring of
    # x_shift differs from sampling x beyond the ring -> compare the interior that does not touch it
    assert torch.allclose(got[:, :, 1:-1, 1:-1], want[:, :, 1:-1, 1:-1], atol=1e-5)


def test_fractional_offsets_interpolate():
    # single channel, 1x1 kernel, weight 1: the op IS bilinear sampling
    x = torch.arange(25, dtype=torch.float32).view(1, 1, 5, 5)
    off = torch.zeros(1, 2, 5, 5)
    off[:, 0], off[:, 1] = 0.5, 0.25
    got = dcn_oracle.deform_conv(x, off, torch.ones(1, 1, 1, 1), 1, 0, 1, 1, 1)
    # interior: value + 0.5*5 + 0.25*1
    assert torch.allclose(got[0, 0, :4, :4], x[0, 0, :4, :4] + 2.75, atol=1e-5)
    # last row samples h=4.5: rows 4 and 5(out of range -> 0): half weight
    assert torch.allclose(got[0, 0, 4, :4], 0.5 * (x[0, 0, 4, :4] * 0.75 + x[0, 0, 4, 1:5] * 0.25), atol=1e-5)


def test_mask_one_equals_v1_plus_bias_and_mask_scales():
    x, w, b = rnd((2, 6, 9, 9), 5), rnd((4, 6, 3, 3), 6, 0.2), rnd((4,), 7)
    off = rnd((2, 2 * 2 * 9, 9, 9), 8, 1.5)                      # deformable_groups = 2
    v1 = dcn_oracle.deform_conv(x, off, w, 1, 1, 1, 1, 2)
    v2 = dcn_oracle.deform_conv(x, off, w, 1, 1, 1, 1, 2, mask=torch.ones(2, 2 * 9, 9, 9), bias=b)
    assert torch.allclose(v2, v1 + b.view(1, -1, 1, 1), atol=1e-5)
    half = dcn_oracle.deform_conv(x, off, w, 1, 1, 1, 1, 2, mask=torch.full((2, 18, 9, 9), 0.5))
    assert torch.allclose(half, 0.5 * v1, atol=1e-5)


def test_deformable_groups_use_their_own_offsets():
    x, w = rnd((1, 4, 8, 8), 9), rnd((3, 4, 3, 3), 10, 0.3)
    off = torch.zeros(1, 2 * 18, 8, 8)
    off[:, 18:] = 1.0                                            # second group (channels 2,3) shifted by (+1,+1)
    got = dcn_oracle.deform_conv(x, off, w, 1, 1, 1, 1, 2)
    a = dcn_oracle.deform_conv(x[:, :2], torch.zeros(1, 18, 8, 8), w[:, :2].contiguous(), 1, 1, 1, 1, 1)
    b = dcn_oracle.deform_conv(x[:, 2:], torch.ones(1, 18, 8, 8), w[:, 2:].contiguous(), 1, 1, 1, 1, 1)
    assert torch.allclose(got, a + b, atol=1e-5)


def test_invalid_shapes_rejected():
    with pytest.raises(ValueError):
        dcn_oracle.deform_conv(torch.zeros(1, 3, 2, 2), torch.zeros(1, 18, 1, 1), torch.zeros(2, 3, 3, 3), 1, 0, 1, 1, 1)


# ---- backward (oracle/dcn_oracle.c: dcn_oracle_backward) -------------------------------------------------------------------------------

@pytest.mark.parametrize("k,stride,pad,dil,groups", [(3, 1, 1, 1, 1), (3, 2, 1, 1, 1), (3, 1, 2, 2, 2), (1, 1, 0, 1, 1)])
def test_backward_at_zero_offsets_equals_conv2d_autograd(k, stride, pad, dil, groups):
    x, w, b = rnd((2, 8, 9, 10), 1).requires_grad_(), rnd((6, 8 // groups, k, k), 2, 0.2).requires_grad_(), rnd((6,), 3).requires_grad_()
    y = F.conv2d(x, w, b, stride, pad, dil, groups)
    gy = rnd(tuple(y.shape), 4)
    y.backward(gy)
    off = torch.zeros(2, 2 * k * k, y.shape[2], y.shape[3])
    mask = torch.ones(2, k * k, y.shape[2], y.shape[3])
    for m in (None, mask):
        g = dcn_oracle.deform_conv_backward(x.detach(), off, w.detach(), gy, stride, pad, dil, groups, 1, mask=m, with_bias=True)
        assert torch.allclose(g["input"], x.grad, atol=2e-5)
        assert torch.allclose(g["weight"], w.grad, atol=2e-4)
        assert torch.allclose(g["bias"], b.grad, atol=2e-4)


@pytest.mark.parametrize("dg,groups,with_mask", [(1, 1, False), (2, 1, True), (4, 2, True)])
def test_backward_matches_finite_differences_of_the_forward_oracle(dg, groups, with_mask):
    """d loss / d offset, d loss / d mask, d loss / d input, d loss / d weight of loss = <forward, grad_out>, against central differences of
    dcn_oracle.deform_conv (itself pinned above).  Offsets are kept away from integer sample positions, where the bilinear sample has a kink."""
    B, C, H, W, Cout, k = 1, 8, 7, 6, 4, 3
    x, w = rnd((B, C, H, W), 11), rnd((Cout, C // groups, k, k), 12, 0.3)
    g = torch.Generator().manual_seed(13)
    off = (torch.rand(B, dg * 2 * k * k, H, W, generator=g) * 0.6 + 0.2) * (torch.randint(0, 2, (B, dg * 2 * k * k, H, W), generator=g) * 2 - 1) * 1.3
    frac = off - off.floor()
    off = torch.where((frac < 0.1) | (frac > 0.9), off.floor() + 0.5, off)
    mask = torch.rand(B, dg * k * k, H, W, generator=g) if with_mask else None
    gy = rnd((B, Cout, H, W), 14)
    fwd = lambda x_, off_, w_, m_: float((dcn_oracle.deform_conv(x_, off_, w_, 1, 1, 1, groups, dg, mask=m_).double() * gy.double()).sum())
    got = dcn_oracle.deform_conv_backward(x, off, w, gy, 1, 1, 1, groups, dg, mask=mask)
    eps = 1e-2
    picks = torch.randperm(off.numel(), generator=g)[:40]
    for idx in picks.tolist():
        d = torch.zeros(off.numel()); d[idx] = eps
        d = d.view_as(off)
        num = (fwd(x, off + d, w, mask) - fwd(x, off - d, w, mask)) / (2 * eps)
        assert abs(num - float(got["offset"].flatten()[idx])) <= 2e-3 * max(1.0, abs(num)), (idx, num, float(got["offset"].flatten()[idx]))
    for name, t in (("input", x), ("weight", w)) + ((("mask", mask),) if with_mask else ()):
        for idx in torch.randperm(t.numel(), generator=g)[:25].tolist():
            d = torch.zeros(t.numel()); d[idx] = eps
            d = d.view_as(t)
            def at(sign):
                if name == "input":
                    return fwd(x + sign * d, off, w, mask)
                if name == "weight":
                    return fwd(x, off, w + sign * d, mask)
                return fwd(x, off, w, mask + sign * d)
            num = (at(1) - at(-1)) / (2 * eps)                      # the forward is linear in each of these: exact up to fp32 rounding
            assert abs(num - float(got[name].flatten()[idx])) <= 2e-3 * max(1.0, abs(num)), (name, idx)


def test_backward_is_linear_in_grad_output_and_accumulates_weight_grads():
    x, w = rnd((2, 4, 6, 6), 21), rnd((4, 4, 3, 3), 22, 0.3)
    off, mask = rnd((2, 18, 6, 6), 23, 0.7), torch.rand(2, 9, 6, 6, generator=torch.Generator().manual_seed(24))
    g1, g2 = rnd((2, 4, 6, 6), 25), rnd((2, 4, 6, 6), 26)
    a = dcn_oracle.deform_conv_backward(x, off, w, g1, 1, 1, 1, 1, 1, mask=mask, with_bias=True)
    b = dcn_oracle.deform_conv_backward(x, off, w, g2, 1, 1, 1, 1, 1, mask=mask, with_bias=True)
    c = dcn_oracle.deform_conv_backward(x, off, w, g1 * 2 - g2, 1, 1, 1, 1, 1, mask=mask, with_bias=True)
    for k in a:
        assert torch.allclose(c[k], 2 * a[k] - b[k], atol=1e-4), k
    half = dcn_oracle.deform_conv_backward(x, off, w, g1, 1, 1, 1, 1, 1, mask=mask, scale=0.5)
    assert torch.allclose(half["weight"], 0.5 * a["weight"], atol=1e-5)       # deform_conv_backward_parameters_cuda's `scale`
