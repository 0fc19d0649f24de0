"""ctypes binding of libcfen_hip.so (C ABI in include/cfen_hip.h).

There is NO fallback: if the library is missing or a call fails, an exception is raised.  The
product path never routes through PyTorch ops or the CPU oracle.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CFEN_HIP_LIB") or os.path.join(HERE, "libcfen_hip.so")   # override: A/B builds of the same ABI

CFEN_F32, CFEN_F16 = 0, 1
c_void_p, c_int, c_float, c_size_t, c_char_p = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t, ctypes.c_char_p


class CfenError(RuntimeError):
    pass


class NetConfigC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("batch", "n_feats", "hidden_dim_ratio", "patch_size", "load_size", "num_heads", "dtype", "reserved")]


class ConvArgsC(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("kind", "k", "stride", "pad", "reflect", "nsrc", "B", "Hin", "Win", "Cin", "cs_in",
                 "Cout", "Cout_pad", "Kpad", "cs_out", "act", "out_nchw_f32", "cs_res", "wlayout")] + \
               [(n, c_void_p) for n in ("src0", "src1", "weight", "scale", "shift", "res0", "res1", "out", "src2")]


class EmbedQkvArgsC(ctypes.Structure):
    _fields_ = [("fmap", c_void_p)] + [(n, ctypes.c_int32) for n in ("B", "H", "W", "C", "cs", "ws", "p")] + \
               [(n, c_void_p) for n in ("we", "be", "pos", "ln_gamma", "ln_beta", "wqkv", "x1", "qkv")] + [("eps", c_float), ("head_major_heads", ctypes.c_int32)]


class MlpArgsC(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "y", "fmap", "att", "w_proj", "ln_gamma", "ln_beta", "w1a", "b1a", "w2a", "b2a", "w1b", "b1b", "w2b", "b2b")] + \
               [("M", ctypes.c_int64), ("D", ctypes.c_int32), ("H", ctypes.c_int32), ("eps", c_float)] + \
               [(n, ctypes.c_int32) for n in ("mapH", "mapW", "C", "cs", "ws", "p")]


class MlpStreamArgsC(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "y", "fmap", "att", "wp_stream", "ln_gamma", "ln_beta", "wa_stream", "b1a", "b2a", "wb_stream", "b1b", "b2b")] + \
               [("M", ctypes.c_int64), ("D", ctypes.c_int32), ("H", ctypes.c_int32), ("eps", c_float)] + \
               [(n, ctypes.c_int32) for n in ("mapH", "mapW", "C", "cs", "ws", "p")]


class LvitArgsC(ctypes.Structure):
    _fields_ = [("fmap", c_void_p), ("out", c_void_p)] + [(n, ctypes.c_int32) for n in ("B", "H", "W", "C", "cs_in", "cs_out", "ws", "p")] + \
               [(n, c_void_p) for n in ("w_stream", "be", "pos", "ln1_gamma", "ln1_beta", "ln2_gamma", "ln2_beta", "b1a", "b2a", "b1b", "b2b")] + \
               [("hidden", ctypes.c_int32), ("eps", c_float)]


class ChainPhaseC(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in ("x", "w_stream", "bias", "lnf_s", "residual", "pos", "y")] + \
               [(n, ctypes.c_int32) for n in ("ldx", "ldr", "ldy", "period", "N", "K", "relu", "nsplit", "fold")]


class ChainArgsC(ctypes.Structure):
    _fields_ = [("phase", ChainPhaseC * 5), ("nphases", ctypes.c_int32), ("M", ctypes.c_int32)] + \
               [(n, ctypes.c_int32) for n in ("fold_H", "fold_W", "fold_cs", "fold_C", "fold_p")] + \
               [("sync_ws", c_void_p), ("sync_ws_bytes", c_size_t)]


# every symbol include/cfen_hip.h declares: (restype, argtypes)
_I = c_int
_P = c_void_p
SIGNATURES = {
    "cfen_abi_version": (_I, []),
    "cfen_last_error": (c_char_p, []),
    "cfen_tune": (_I, [c_char_p, _I]),
    "cfen_net_create": (_I, [ctypes.POINTER(_P), ctypes.POINTER(NetConfigC)]),
    "cfen_net_destroy": (None, [_P]),
    "cfen_net_workspace_bytes": (c_size_t, [_P]),
    "cfen_net_set_param": (_I, [_P, c_char_p, _P, c_size_t]),
    "cfen_net_missing_params": (_I, [_P, c_char_p, c_size_t]),
    "cfen_net_actnorm_pending": (_I, [_P, c_char_p, _P, _P, _P]),
    "cfen_net_actnorm_pending_count": (_I, [_P]),
    "cfen_net_set_input_u8": (_I, [_P, _I]),
    "cfen_net_set_output_u8": (_I, [_P, _I]),
    "cfen_net_set_output_f16": (_I, [_P, _I]),
    "cfen_net_forward": (_I, [_P, _P, _P, _P, _P, _P, c_size_t, _P]),
    "cfen_net_graph_capture": (_I, [_P, _P, _P, _P, _P, _P, c_size_t, ctypes.POINTER(ctypes.c_int32)]),
    "cfen_net_graph_launch": (_I, [_P, ctypes.c_int32, _P]),
    "cfen_net_profile": (_I, [_P, _P, _P, _P, _P, _P, c_size_t, _P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                              ctypes.POINTER(ctypes.c_int32), _I]),
    "cfen_net_profile_entry": (_I, [_P, _I, ctypes.POINTER(c_char_p), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(ctypes.c_double),
                                    ctypes.POINTER(ctypes.c_double)]),
    "cfen_net_profile_entry_kernel": (_I, [_P, _I, ctypes.POINTER(c_char_p), ctypes.POINTER(ctypes.c_double)]),
    "cfen_net_stage": (_I, [_P, c_char_p, ctypes.POINTER(_P)] + [ctypes.POINTER(ctypes.c_int32)] * 4),
    "cfen_net_flops_per_image": (ctypes.c_double, [_P]),
    "cfen_net_chain_error_words": (_I, [_P, ctypes.POINTER(_P), _I]),
    "cfen_gemm_nt": (_I, [_I, _P, _I, _P, _I, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _P]),
    "cfen_gemm_ln": (_I, [_I, _P, _I, _P, _I, _P, _P, _P, _I, _I, _I, _I, _I, ctypes.c_float, _P]),
    "cfen_gemm_splitk": (_I, [_I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, c_size_t, _P]),
    "cfen_head_conv5": (_I, [_I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "cfen_gemm_chain": (_I, [_I, ctypes.POINTER(ChainArgsC), _I, _P]),
    "cfen_embed_gather": (_I, [_I, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P, _I, _P]),
    "cfen_u8hwc_to_nhwc": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "cfen_tensor2im_u8": (_I, [_P, _P, _I, _I, _I, _P]),
    "cfen_embed_qkv": (_I, [_I, ctypes.POINTER(EmbedQkvArgsC), _P]),
    "cfen_embed_qkv_stream": (_I, [_I, ctypes.POINTER(EmbedQkvArgsC), _P]),
    "cfen_layernorm": (_I, [_I, _P, _P, _P, _P, _I, _I, c_float, _P]),
    "cfen_attention": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "cfen_attention_head_major": (_I, [_I, _P, _P, _I, _I, _I, _I, _P]),
    "cfen_mlp_block": (_I, [_I, ctypes.POINTER(MlpArgsC), _P]),
    "cfen_mlp_stream_block": (_I, [_I, ctypes.POINTER(MlpStreamArgsC), _P]),
    "cfen_lvit_window": (_I, [_I, ctypes.POINTER(LvitArgsC), _P]),
    "cfen_patchify": (_I, [_I, _P, _P] + [_I] * 8 + [_P]),
    "cfen_unpatchify": (_I, [_I, _P, _P] + [_I] * 7 + [_P]),
    "cfen_upsample4": (_I, [_I, _P, _P] + [_I] * 6 + [_P]),
    "cfen_nchw_to_nhwc": (_I, [_I, _P, _P] + [_I] * 5 + [_P]),
    "cfen_conv2d": (_I, [_I, ctypes.POINTER(ConvArgsC), _P]),
    "cfen_stats_workspace": (c_size_t, [_I, _I]),
    "cfen_instnorm_relu": (_I, [_I, _P, _P, _I, _I, _I, _I, c_float, _P]),
    "cfen_cfsm2g": (_I, [_I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "cfen_deform_conv_columns_bytes": (c_size_t, [_I] * 9),
    "cfen_deform_conv_forward": (_I, [_I, _P, _P, _P, _P] + [_I] * 16 + [_P, c_size_t, _P]),
    "cfen_modulated_deform_conv_forward": (_I, [_I, _P, _P, _P, _P, _P, _P] + [_I] * 16 + [_P, c_size_t, _P]),
    "cfen_deform_conv_forward_nhwc": (_I, [_I, _P, _P, _P, _P] + [_I] * 16 + [_P, c_size_t, _P]),
    "cfen_modulated_deform_conv_forward_nhwc": (_I, [_I, _P, _P, _P, _P, _P, _P] + [_I] * 16 + [_P, c_size_t, _P]),
    "cfen_deform_conv_backward_bytes": (c_size_t, [_I] * 10),
    "cfen_deform_conv_backward_set_lds": (_I, [_I]),
    "cfen_deform_conv_backward_input": (_I, [_I, _P, _P, _P, _P, _P, _P] + [_I] * 16 + [_P, c_size_t, _P]),
    "cfen_deform_conv_backward_parameters": (_I, [_I, _P, _P, _P, _P] + [_I] * 15 + [c_float, _I, _P, c_size_t, _P]),
    "cfen_modulated_deform_conv_backward": (_I, [_I] + [_P] * 11 + [_I] * 16 + [_P, c_size_t, _P]),
}

_lib = None


def load():
    """Load the shared library once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcfen_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950) -- there is no CPU/PyTorch fallback for the HIP path")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().cfen_last_error()
        raise CfenError("%s failed (%d): %s" % (what or "cfen call", rc, msg.decode() if msg else "?"))


def dtype_code(torch_dtype):
    import torch
    if torch_dtype == torch.float16:
        return CFEN_F16
    if torch_dtype == torch.float32:
        return CFEN_F32
    raise TypeError("HIP path supports float16 and float32 storage, got %s" % torch_dtype)


def ptr(t):
    return c_void_p(t.data_ptr()) if t is not None else c_void_p(0)


def current_stream():
    import torch
    return c_void_p(torch.cuda.current_stream().cuda_stream)
