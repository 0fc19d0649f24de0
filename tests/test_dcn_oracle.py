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
