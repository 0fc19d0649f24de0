"""Deformable convolution operator with the reference's Python API, backed by the HIP kernel.

Mirrors dcn/deform_conv.py of the reference (functions :15-154, modules :161-329): same call signatures,
argument meaning, error behaviour (ValueError for non-4D input and too-small outputs, AssertionError when
im2col_step does not divide the batch, NotImplementedError for CPU tensors).  The pybind module
`deform_conv_cuda` (dcn/src/deform_conv_cuda.cpp:681-695) is replaced by C-ABI entry points of
libcfen_hip.so: forward (csrc/k_dcn.hip; the column matrix is never materialised, the `columns` tensor is
the kernel's operand scratch) and backward (csrc/k_dcn_bwd.hip: grad_input / grad_offset / grad_mask /
grad_weight / grad_bias, fp32 arithmetic), so the functions are differentiable like the reference's.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from .. import _lib
from .._lib import check, ptr, dtype_code, current_stream


def _contig(*ts):
    return [t.contiguous() if t is not None else None for t in ts]


def _nhwc_input(ctx, input, weight, groups):
    """(extension, round 6) a channels_last input in a forward nobody differentiates goes to the kernel as it is: the NHWC copy the kernel samples from IS its memory
    (cfen_*_forward_nhwc).  The backward kernels read NCHW, so a differentiated call keeps the reference's contiguous() copy."""
    ve = 16 // input.element_size()
    return (input.dim() == 4 and not any(ctx.needs_input_grad) and not input.is_contiguous() and input.is_contiguous(memory_format=torch.channels_last)
            and input.dtype in (torch.float16, torch.float32) and (input.size(1) // groups) % ve == 0 and input.size(1) % groups == 0)


def _columns(input, weight, groups):
    """The reference allocates `columns` / `ones` scratch tensors per call (deform_conv.py:31, 106-107) and hands them to the extension;
    here `columns` is the scratch of the HIP kernel's NHWC / tap-major operand copies (csrc/k_dcn.hip).  (buffer, nbytes)."""
    B, C, H, W = input.shape
    n = _lib.load().cfen_deform_conv_columns_bytes(dtype_code(input.dtype), B, C, H, W, weight.size(0), weight.size(2), weight.size(3), groups)
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=input.device), int(n)


def _backward_scratch(input, weight, out_hw, groups):
    B, C, H, W = input.shape
    n = _lib.load().cfen_deform_conv_backward_bytes(B, C, H, W, weight.size(0), weight.size(2), weight.size(3), out_hw[0], out_hw[1], groups)
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=input.device), int(n)


class DeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, im2col_step=64):
        if input is not None and input.dim() != 4:
            raise ValueError("Expected 4D tensor as input, got {}D tensor instead.".format(input.dim()))
        stride, padding, dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.stride, ctx.padding, ctx.dilation = stride, padding, dilation
        ctx.groups, ctx.deformable_groups, ctx.im2col_step = groups, deformable_groups, im2col_step
        output = input.new_empty(DeformConvFunction._output_size(input, weight, padding, dilation, stride))
        if not input.is_cuda:
            raise NotImplementedError
        cur_im2col_step = min(im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        if offset.shape[1] != deformable_groups * 2 * weight.size(2) * weight.size(3):
            raise RuntimeError("invalid number of channels of offset")
        if tuple(offset.shape[2:]) != tuple(output.shape[2:]):
            raise RuntimeError("invalid spatial size of offset, expected height: %d width: %d, but got height: %d width: %d"
                               % (output.shape[2], output.shape[3], offset.shape[2], offset.shape[3]))
        if input.size(1) != weight.size(1) * groups:
            raise RuntimeError("invalid number of input planes, expected: %d, but got: %d" % (weight.size(1) * groups, input.size(1)))
        nhwc = _nhwc_input(ctx, input, weight, groups)
        offset, weight = _contig(offset.to(input.dtype), weight.to(input.dtype))
        if not nhwc:
            input = input.contiguous()
        ctx.save_for_backward(input, offset, weight)
        B, C, H, W = input.shape
        columns, nbytes = _columns(input, weight, groups)
        lib = _lib.load()
        # note the reference passes W before H here (deform_conv.py:41-46)
        check((lib.cfen_deform_conv_forward_nhwc if nhwc else lib.cfen_deform_conv_forward)(
            dtype_code(input.dtype), ptr(input), ptr(weight), ptr(offset), ptr(output), B, C, H, W, weight.size(0),
            weight.size(3), weight.size(2), stride[1], stride[0], padding[1], padding[0], dilation[1], dilation[0],
            groups, deformable_groups, cur_im2col_step, ptr(columns), nbytes, current_stream()), "deform_conv_forward")
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        """dcn/deform_conv.py:49-80: grad_input + grad_offset when either is needed, grad_weight when needed"""
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        if not grad_output.is_cuda:
            raise NotImplementedError
        cur_im2col_step = min(ctx.im2col_step, input.shape[0])
        assert (input.shape[0] % cur_im2col_step) == 0, 'im2col step must divide batchsize'
        grad_output = grad_output.to(input.dtype).contiguous()
        B, C, H, W = input.shape
        geom = (B, C, H, W, weight.size(0), weight.size(3), weight.size(2), ctx.stride[1], ctx.stride[0], ctx.padding[1], ctx.padding[0],
                ctx.dilation[1], ctx.dilation[0], ctx.groups, ctx.deformable_groups)
        columns, nbytes = _backward_scratch(input, weight, grad_output.shape[2:], ctx.groups)
        lib, dt = _lib.load(), dtype_code(input.dtype)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            check(lib.cfen_deform_conv_backward_input(dt, ptr(input), ptr(offset), ptr(grad_output), ptr(grad_input), ptr(grad_offset), ptr(weight),
                                                      *geom, cur_im2col_step, ptr(columns), nbytes, current_stream()), "deform_conv_backward_input")
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            check(lib.cfen_deform_conv_backward_parameters(dt, ptr(input), ptr(offset), ptr(grad_output), ptr(grad_weight), *geom, 1.0,
                                                           cur_im2col_step, ptr(columns), nbytes, current_stream()), "deform_conv_backward_parameters")
        return (grad_input, grad_offset, grad_weight, None, None, None, None, None, None)

    @staticmethod
    def _output_size(input, weight, padding, dilation, stride):
        channels = weight.size(0)
        output_size = (input.size(0), channels)
        for d in range(input.dim() - 2):
            in_size = input.size(d + 2)
            pad = padding[d]
            kernel = dilation[d] * (weight.size(d + 2) - 1) + 1
            stride_ = stride[d]
            output_size += ((in_size + (2 * pad) - kernel) // stride_ + 1, )
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError("convolution input is too small (output would be {})".format('x'.join(map(str, output_size))))
        return output_size


class ModulatedDeformConvFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1):
        with_bias = bias is not None
        if not input.is_cuda:
            raise NotImplementedError
        n, channels_out = input.size(0), weight.size(0)
        height, width = input.shape[2:4]
        kernel_h, kernel_w = weight.shape[2:4]
        height_out = (height + 2 * padding - (dilation * (kernel_h - 1) + 1)) // stride + 1
        width_out = (width + 2 * padding - (dilation * (kernel_w - 1) + 1)) // stride + 1
        output = input.new_empty((n, channels_out, height_out, width_out))
        if input.size(1) != weight.size(1) * groups:
            raise RuntimeError("Input shape and kernel channels wont match: (%d vs %d)." % (input.size(1), weight.size(1) * groups))
        nhwc = _nhwc_input(ctx, input, weight, groups)
        offset, mask, weight, bias = _contig(offset.to(input.dtype), mask.to(input.dtype), weight.to(input.dtype), bias.to(input.dtype) if with_bias else None)
        if not nhwc:
            input = input.contiguous()
        ctx.stride, ctx.padding, ctx.dilation, ctx.groups, ctx.deformable_groups, ctx.with_bias = stride, padding, dilation, groups, deformable_groups, with_bias
        ctx.save_for_backward(input, offset, mask, weight)
        columns, nbytes = _columns(input, weight, groups)
        # scalar stride / padding / dilation, h before w (deform_conv.py:117-119)
        lib = _lib.load()
        check((lib.cfen_modulated_deform_conv_forward_nhwc if nhwc else lib.cfen_modulated_deform_conv_forward)(
            dtype_code(input.dtype), ptr(input), ptr(weight), ptr(bias), ptr(offset), ptr(mask), ptr(output), n, input.size(1),
            height, width, channels_out, kernel_h, kernel_w, stride, stride, padding, padding, dilation, dilation, groups,
            deformable_groups, int(with_bias), ptr(columns), nbytes, current_stream()), "modulated_deform_conv_forward")
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        """dcn/deform_conv.py:122-145: all five gradients in one extension call"""
        if not grad_output.is_cuda:
            raise NotImplementedError
        input, offset, mask, weight = ctx.saved_tensors
        grad_output = grad_output.to(input.dtype).contiguous()
        grad_input, grad_offset, grad_mask, grad_weight = (torch.zeros_like(t) for t in (input, offset, mask, weight))
        grad_bias = torch.zeros(weight.size(0), dtype=input.dtype, device=input.device) if ctx.with_bias else None
        B, C, H, W = input.shape
        columns, nbytes = _backward_scratch(input, weight, grad_output.shape[2:], ctx.groups)
        check(_lib.load().cfen_modulated_deform_conv_backward(
            dtype_code(input.dtype), ptr(input), ptr(weight), None, ptr(offset), ptr(mask), ptr(grad_input), ptr(grad_weight), ptr(grad_bias),
            ptr(grad_offset), ptr(grad_mask), ptr(grad_output), B, C, H, W, weight.size(0), weight.size(2), weight.size(3), ctx.stride, ctx.stride,
            ctx.padding, ctx.padding, ctx.dilation, ctx.dilation, ctx.groups, ctx.deformable_groups, int(ctx.with_bias), ptr(columns), nbytes,
            current_stream()), "modulated_deform_conv_backward")
        return (grad_input, grad_offset, grad_mask, grad_weight, grad_bias, None, None, None, None, None)


deform_conv = DeformConvFunction.apply
modulated_deform_conv = ModulatedDeformConvFunction.apply


# ---- the module API (reference dcn/deform_conv.py:161-329: six nn.Modules) ------------------------------------------------------------------
# One base class owns the geometry, the kernel weight (and v2's bias) and their initialisation; the reference's classes are thin views of it.  What a
# checkpoint or a caller can see is kept: class names, constructor signatures, attribute names, state_dict keys (`weight`, `bias`,
# `conv_offset.*`, `conv_offset_mask.*`), uniform(+-1/sqrt(Cin*kh*kw)) weights, zero bias, zero-initialised offset branches (so a *Pack module is a
# plain convolution until trained), `chunk -> cat -> sigmoid` of the v2 branch's output, the `[input, features]` calling form of extra_offset_mask.

class _DeformableBase(nn.Module):
    _modulated = False          # v2: mask + optional bias, and stride / padding / dilation kept as the caller passed them (the reference does, :221-237)

    def _setup(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias):
        for what, c in (("in_channels", in_channels), ("out_channels", out_channels)):
            assert c % groups == 0, '{} {} cannot be divisible by groups {}'.format(what, c, groups)
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size = _pair(kernel_size)
        keep = (lambda v: v) if self._modulated else _pair
        self.stride, self.padding, self.dilation = keep(stride), keep(padding), keep(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        if self._modulated:
            self.with_bias = bias
            if bias:
                self.bias = nn.Parameter(torch.empty(out_channels))
            else:
                self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels * self.kernel_size[0] * self.kernel_size[1])
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            if getattr(self, 'bias', None) is not None:
                self.bias.zero_()

    def _offset_branch(self, maps_per_tap, in_channels=None):
        """the convolution that predicts offsets (2 maps per tap and deformable group) or offsets + masks (3): same window as the deformable conv, zero at init"""
        taps = self.kernel_size[0] * self.kernel_size[1]
        conv = nn.Conv2d(in_channels or self.in_channels, self.deformable_groups * maps_per_tap * taps, kernel_size=self.kernel_size,
                         stride=_pair(self.stride), padding=_pair(self.padding), bias=True)
        nn.init.zeros_(conv.weight)
        nn.init.zeros_(conv.bias)
        return conv

    def _apply_v1(self, x, offset):
        return deform_conv(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)

    def _apply_v2(self, x, offset, mask):
        return modulated_deform_conv(x, offset, mask, self.weight, self.bias, self.stride, self.padding, self.dilation, self.groups, self.deformable_groups)


class DeformConv(_DeformableBase):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=False):
        super().__init__()
        assert not bias                                    # DCNv1 has no bias (reference :165)
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, False)

    def forward(self, x, offset):
        return self._apply_v1(x, offset)


class DeformConvPack(DeformConv):
    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = self._offset_branch(2)

    def init_offset(self):
        nn.init.zeros_(self.conv_offset.weight)
        nn.init.zeros_(self.conv_offset.bias)

    def forward(self, x):
        return self._apply_v1(x, self.conv_offset(x))


class ModulatedDeformConv(_DeformableBase):
    _modulated = True

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, deformable_groups=1, bias=True):
        super().__init__()
        self._setup(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias)

    def forward(self, x, offset, mask):
        return self._apply_v2(x, offset, mask)


class ModulatedDeformConvPack(ModulatedDeformConv):
    _offset_in_channels = None

    def __init__(self, *args, extra_offset_mask=False, **kwargs):
        super().__init__(*args, **kwargs)
        self.extra_offset_mask = extra_offset_mask
        self.conv_offset_mask = self._offset_branch(3, self._offset_in_channels)

    def init_offset(self):
        nn.init.zeros_(self.conv_offset_mask.weight)
        nn.init.zeros_(self.conv_offset_mask.bias)

    def forward(self, x):
        x, feat = (x[0], x[1]) if self.extra_offset_mask else (x, x)       # extra_offset_mask: x = [input, features the branch reads]
        dy, dx, m = torch.chunk(self.conv_offset_mask(feat), 3, dim=1)
        return self._apply_v2(x, torch.cat((dy, dx), dim=1), torch.sigmoid(m))


class ModulatedDeformConvPack2(ModulatedDeformConvPack):
    """ModulatedDeformConvPack whose offset / mask branch reads `offset_in_channel` feature channels (reference :294-329)."""

    def __init__(self, *args, extra_offset_mask=False, offset_in_channel=32, **kwargs):
        self._offset_in_channels = offset_in_channel
        super().__init__(*args, extra_offset_mask=extra_offset_mask, **kwargs)
