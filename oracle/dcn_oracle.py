"""ctypes handle on oracle/libdcn_oracle.so (C restatement of the deformable-conv forward).
TEST INFRASTRUCTURE ONLY -- see dcn_oracle.c for what it follows and how it is pinned."""
import ctypes
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libdcn_oracle.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(HERE, "dcn_oracle.c")):
            subprocess.check_call(["make", "-s", "-C", HERE])
        _lib = ctypes.CDLL(so)
        _lib.dcn_oracle_forward.restype = ctypes.c_int
        _lib.dcn_oracle_forward.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 15
        _lib.dcn_oracle_backward.restype = ctypes.c_int
        _lib.dcn_oracle_backward.argtypes = [ctypes.c_void_p] * 10 + [ctypes.c_float] + [ctypes.c_int] * 15
    return _lib


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_conv(x, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, mask=None, bias=None):
    """CPU float32 tensors (NCHW) -> output tensor; mask=None is DCNv1."""
    x, offset, weight = x.float().contiguous(), offset.float().contiguous(), weight.float().contiguous()
    mask = mask.float().contiguous() if mask is not None else None
    bias = bias.float().contiguous() if bias is not None else None
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    out = torch.empty(B, Cout, max(Ho, 0), max(Wo, 0))
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = _load().dcn_oracle_forward(p(x), p(offset), p(mask), p(weight), p(bias), p(out), B, C, H, W, Cout, kh, kw, sh, sw, ph, pw,
                                    dh, dw, groups, deformable_groups)
    if rc != 0:
        raise ValueError("dcn_oracle: invalid shape")
    return out


def deform_conv_backward(x, offset, weight, grad_out, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, mask=None,
                         with_bias=False, scale=1.0):
    """CPU float32 tensors -> dict(input, offset, mask, weight, bias) of gradients (mask / bias entries None when absent): the
    reference's deform_conv_backward_input_cuda + deform_conv_backward_parameters_cuda (DCNv1) or modulated_deform_conv_cuda_backward."""
    x, offset, weight, grad_out = (t.float().contiguous() for t in (x, offset, weight, grad_out))
    mask = mask.float().contiguous() if mask is not None else None
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    B, C, H, W = x.shape
    Cout, _, kh, kw = weight.shape
    g = {"input": torch.zeros_like(x), "offset": torch.zeros_like(offset), "mask": torch.zeros_like(mask) if mask is not None else None,
         "weight": torch.zeros_like(weight), "bias": torch.zeros(Cout) if with_bias else None}
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    rc = _load().dcn_oracle_backward(p(x), p(offset), p(mask), p(weight), p(grad_out), p(g["input"]), p(g["offset"]), p(g["mask"]),
                                     p(g["weight"]), p(g["bias"]), scale, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, groups,
                                     deformable_groups)
    if rc != 0:
        raise ValueError("dcn_oracle: invalid shape")
    return g
