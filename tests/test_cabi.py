"""The C-ABI library loads and exports every symbol include/cfen_hip.h declares (no compute calls: runs
without a GPU), and the ctypes signature table covers exactly that set."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cfen_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cfen_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from cfen_vit_dehazing_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(lib, s), "libcfen_hip.so does not export %s" % s
    assert sorted(_lib.SIGNATURES) == syms
    assert _lib.load().cfen_abi_version() == 1


def test_argument_errors_do_not_need_a_gpu():
    from cfen_vit_dehazing_amd import _lib
    lib = _lib.load()
    cfg = _lib.NetConfigC(batch=1, n_feats=24, hidden_dim_ratio=4, patch_size=32, load_size=250, num_heads=4, dtype=1, reserved=0)
    h = ctypes.c_void_p()
    assert lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cfg)) == -1          # loadSize != 8*patch_size
    assert b"loadSize" in lib.cfen_last_error()
    cfg.load_size = 256
    assert lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cfg)) == 0
    assert lib.cfen_net_workspace_bytes(h) > 0
    assert abs(lib.cfen_net_flops_per_image(h) / 1e9 - 120.85) < 0.01           # SURVEY 8d closed form
    assert lib.cfen_net_set_param(h, b"no.such.param", ctypes.c_void_p(16), 4) == -1
    buf = ctypes.create_string_buffer(1 << 16)
    assert lib.cfen_net_missing_params(h, buf, 1 << 16) == 690   # (+ 1: head.0.0.w5, the 8-byte-pixel layout of k_head5; + 4 level-1 GViT blocks x 5 fragment-stream layouts, round 4) (+ 4 level-3 LViT blocks x 5 and 4 level-2 blocks x 3 fragment-stream layouts, round 3; the window kernel's three layouts became one stream) 24 transformer blocks x 21 + 15 conv layers x 3 ... + 8 LViT blocks with the extra fused-front layouts (2 each) + 4 level-1 LViT blocks x 1 window-kernel weight stream + 16 unfused blocks (GViT, LViT-3) x 6 LayerNorm-folded entries
    lib.cfen_net_destroy(h)
    for hdr, gf in ((2, 85.07),):
        cfg.hidden_dim_ratio = hdr
        assert lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cfg)) == 0
        assert abs(lib.cfen_net_flops_per_image(h) / 1e9 - gf) < 0.01
        lib.cfen_net_destroy(h)
    cfg.hidden_dim_ratio, cfg.patch_size, cfg.load_size = 4, 64, 512
    assert lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cfg)) == 0
    assert abs(lib.cfen_net_flops_per_image(h) / 1e9 - 624.21) < 0.01
    lib.cfen_net_destroy(h)


def test_tuning_knobs_validate_without_a_gpu():
    from cfen_vit_dehazing_amd import _lib
    lib = _lib.load()
    assert lib.cfen_tune(b"no.such.knob", 1) == -1 and b"unknown key" in lib.cfen_last_error()
    assert lib.cfen_tune(b"gemm.kernel", 99) == -1
    assert lib.cfen_tune(b"gemm.large", 1) == -1            # tile ids are 2..5 (+10 / +20 for deeper rings)
    assert lib.cfen_tune(b"net.tail_fused", 3) == -1 and lib.cfen_tune(b"tail.segments", 0) == -1 and lib.cfen_tune(b"gemm.mid", 7) == -1
    for key, val in ((b"gemm.kernel", -1), (b"gemm.large", 4), (b"gemm.small", 15), (b"gemm.mid", 2), (b"gemm.splitk", 0), (b"mlp.small_tiles", 10),
                     (b"net.attn_head_major", 1), (b"net.embed_gather", 1), (b"net.fused_front_max_dim", 192), (b"net.skip_classes", 0),
                     (b"net.tail_fused", 2), (b"tail.segments", 1), (b"tail.balance", 0), (b"tail.debug", 0), (b"net.skip_from", -1), (b"net.skip_to", -1),
                     (b"net.extra_launches", 0), (b"net.gvit_dummy_levels", 0)):
        assert lib.cfen_tune(key, val) == 0, key       # (the shipped defaults: the knobs are process-wide)


def _pack_and_register(cfg, dtype, wtile):
    import torch
    from cfen_vit_dehazing_amd import _lib
    from cfen_vit_dehazing_amd.hipnet import _VARIANT_CODE
    from cfen_vit_dehazing_amd.manifest import generate_state_dict
    from cfen_vit_dehazing_amd.packing import pack_state_dict
    lib = _lib.load()
    td = torch.float16 if dtype == "fp16" else torch.float32
    packed = pack_state_dict(generate_state_dict(cfg, seed=0, with_dead=False), cfg, td, wtile=wtile)
    cc = _lib.NetConfigC(batch=2, n_feats=cfg.n_feats, hidden_dim_ratio=cfg.hidden_dim_ratio, patch_size=cfg.patch_size, load_size=cfg.load_size,
                         num_heads=cfg.num_heads, dtype=_lib.dtype_code(td), reserved=(_VARIANT_CODE[cfg.variant] << 8) | (2 if wtile else 0))
    h = ctypes.c_void_p()
    assert lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cc)) == 0, lib.cfen_last_error()
    keep = {}
    for name, t in packed.items():
        if isinstance(t, str):
            t = packed[t[1:]]
        keep[name] = t = t.contiguous()
        assert lib.cfen_net_set_param(h, name.encode(), ctypes.c_void_p(t.data_ptr()), t.numel() * t.element_size()) == 0, lib.cfen_last_error()
    buf = ctypes.create_string_buffer(1 << 16)
    assert lib.cfen_net_missing_params(h, buf, 1 << 16) == 0, buf.value[:400]
    lib.cfen_net_destroy(h)


@pytest.mark.parametrize("variant,wtile", [("v3", False), ("v3", True), ("cfs", False), ("crs", False), ("v5", False), ("v5", True)])
@pytest.mark.parametrize("dtype", ["fp16", "fp32"])
def test_packed_parameters_are_exactly_what_the_launch_plan_asks_for(variant, wtile, dtype):
    """packing.pack_state_dict (host) and cfen_net::build (csrc/cfen_net.cpp) must agree on every packed name and byte size, for each of
    the four generators; set_param only records the pointer, so this runs without a GPU"""
    from cfen_vit_dehazing_amd.config import NetConfig
    _pack_and_register(NetConfig(24, 4, patch_size=8, load_size=64, variant=variant), dtype, wtile)


@pytest.mark.parametrize("n_feats,hdr,patch,dtype", [(24, 3, 32, "fp16"), (24, 1, 32, "fp16"), (8, 4, 8, "fp16"), (8, 4, 8, "fp32"), (16, 3, 8, "fp16"),
                                                    (32, 2, 8, "fp16")])
def test_packed_parameters_match_the_plan_where_the_fused_kernels_do_not_apply(n_feats, hdr, patch, dtype):
    """the host predicates (packing.window_fusable / mlp_is_fused / front_is_fused / mlp_is_streamed, the LayerNorm fold's 128-byte rule)
    mirror cfen_net::build also off the benchmarked shape: odd hidden_dim_ratio at the window kernel's geometry (hidden % 64 != 0),
    embedding dims whose rows are not whole 128-byte K steps (n_feats 8: D = 32, 128), and the D = 384 fragment-stream blocks"""
    from cfen_vit_dehazing_amd.config import NetConfig
    _pack_and_register(NetConfig(n_feats, hdr, patch_size=patch, load_size=8 * patch), dtype, True)
