"""`dec_ipt` -- the v3 generator as an nn.Module whose forward is the HIP launch plan.

Drop-in for the module `define_G(opt, conv)` returns in the reference
(models/networks_iid_hlgvit_crs_gd4_cfs_v3.py:93-100, class dec_ipt v3:103-1023):
  * same 958-key state_dict (strict load of real `<epoch>_net_G.pth` files works, including the
    208.6 M never-used `decoder.*` / `query_embed` parameters, which are kept on the host side only);
  * `net(x)` with x (B,3,H,W) float32 in [-1,1] returns [xr (B,3,H,W), xs (B,1,H,W), xd (B,3,H,W)] float32.
The arithmetic runs entirely in libcfen_hip.so (csrc/cfen_net.cpp); this class owns parameters,
packed device copies, the workspace and one C `cfen_net` per batch size.
"""
import ctypes
from ctypes import c_char_p as c_char_p_
import os

import torch
from torch import nn

from . import _lib
from ._lib import CfenError, NetConfigC, check, ptr, current_stream
from .config import NetConfig, config_from_opt
from .manifest import state_manifest, _is_dead
from .packing import pack_state_dict

_VARIANT_CODE = {"v3": 0, "cfs": 1, "crs": 2, "v5": 3}       # cfen_net_config.reserved bits 8..15 (csrc/cfen_net.cpp)
_DTYPES = {"fp16": torch.float16, "half": torch.float16, "fp32": torch.float32, "single": torch.float32,
           torch.float16: torch.float16, torch.float32: torch.float32}


class _Node(nn.Module):
    """Anonymous container so dotted reference keys map onto a real module tree."""


def _build_tree(root, manifest):
    """Register every LIVE entry of the manifest as a parameter / buffer under its dotted reference key.  The 208.6 M parameters
    the forward never reads (`decoder.*`, `query_embed`, `sub_mean`, `add_mean`: v3:1158-1168, common.py:16-26) are not
    registered: they would cost 0.83 GB of host memory and ride along on every .to(device).  dec_ipt keeps what a checkpoint
    holds for them in `_dead` (CPU) and hands it back from state_dict()."""
    for key, shape, dt in manifest:
        if _is_dead(key):
            continue
        parts = key.split(".")
        mod = root
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Node())
            mod = getattr(mod, p)
        if dt == torch.int64:
            if key.endswith("position_ids"):
                mod.register_buffer(parts[-1], torch.arange(shape[1]).expand(1, -1).clone())
            else:
                mod.register_buffer(parts[-1], torch.tensor(0))            # ActNorm `initialized` (models/actnorm.py:16)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


class dec_ipt(nn.Module):
    def __init__(self, opt, conv=None, compute_dtype="fp16"):
        super().__init__()
        self.opt = opt
        self.cfg = config_from_opt(opt) if not isinstance(opt, NetConfig) else opt
        self.scale_idx = 0
        self.compute_dtype = _DTYPES[compute_dtype]
        self._manifest = state_manifest(self.cfg)
        _build_tree(self, self._manifest)
        self._dead = {}              # dead key -> CPU tensor, as loaded from a checkpoint (absent = zeros of the manifest shape)
        self._packed = None          # {name: device tensor} of the current compute dtype
        self._packed_cache = {}      # other compute dtypes' packed sets (set_compute_dtype)
        self._packed_dev = None
        self._an_pending = {}        # packed layer name -> (ActNorm key prefix, conv bias, an_out): uninitialised ActNorm2d layers
        self._nets = {}              # (batch, input kind, plan, replica, output kind, compute dtype) -> (handle, workspace tensor)
        self._native_u8 = {}         # same key -> the net writes uint8 outputs itself
        self._graph_keep = []
        self._graphs = []            # capture() handle -> (net key, native graph id)
        self._last = None
        # Launch plan of one forward: two lanes (GViT beside LViT on a second stream / graph branch; best for ONE forward at a time on the default 4
        # hardware queues: 2.75 against 3.44 ms) or one serial chain of launches (what several forwards in flight want -- bench.py, pipeline.py set
        # `serial_plan` themselves -- and what one forward wants once the process runs on more than 4 hardware queues: the two-lane plan's forks / joins
        # then cross queues, 4.49 against 3.44 ms, profiles/r04_ab_hw_queues.txt).  The plan changes WHEN kernels run, never which kernels run or what
        # they compute (round 5: "net.gvit_stream" = 2), so outputs are bitwise the same either way; `plan_info()` says which one is in force and why.
        try:
            many_queues = int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) > 4
        except ValueError:
            many_queues = False
        self.serial_plan = bool(os.environ.get("CFEN_SERIAL")) or many_queues
        self._plan_reason = ("CFEN_SERIAL is set" if os.environ.get("CFEN_SERIAL") else
                             "GPU_MAX_HW_QUEUES=%s > 4 in the environment" % os.environ.get("GPU_MAX_HW_QUEUES") if many_queues else
                             "default: one forward at a time on <= 4 hardware queues")
        # `replica`: which launch plan + workspace the next forward / capture uses.  Replicas share the packed weights; each has its own workspace
        # (stage buffers, token scratch), so forwards of DIFFERENT replicas may be in flight at once on different streams (bench.py --in-flight 2)
        self.replica = 0
        # `output_u8`: forward() returns three uint8 (B,H,W,3) tensors = util.tensor2im of the fp32 results (test.py --out_all saves exactly those),
        # written by the tails' last launch where the geometry allows (cfen_net_set_output_u8), else by cfen_tensor2im_u8 passes on the device
        self.output_u8 = False
        # `output_f16` (round 6): forward() writes / returns fp16 NCHW outputs (the fp32 results rounded to nearest even) straight from the fused tail launch
        # (cfen_net_set_output_f16) -- the sharded run's wire type without a conversion pass.  fp16 nets at full-size tails only: refused loudly elsewhere
        self.output_f16 = False
        # GViT weights tile-major (packing.pack_wtile; cfen_net_config.reserved bit 1): +1 % measured (3.34 -> 3.30 ms at B = 8); CFEN_WTILE=0 = row-major
        self.wtile = os.environ.get("CFEN_WTILE", "1") != "0"
        # CFEN_GVIT_CHAIN=1: GViT weights ALSO as MFMA fragment streams (packing.pack_stream_tiles; cfen_net_config.reserved bit 2) and the GEMMs of a
        # GViT block run as two persistent chains (csrc/k_gvit.hip) instead of eight launches; fp16 only.  OFF by default: measured slower inside the
        # forward (3.15 against 2.80 ms at B = 8, round 4: grid barriers, split-K seams and write-through stores cost more than the launches they replace)
        self.gvit_chain = os.environ.get("CFEN_GVIT_CHAIN", "0") != "0"

    # ---- parameter management ---------------------------------------------------------------
    def state_dict(self, *args, **kw):
        """The reference's 958 keys in the reference's order; dead entries come from the loaded checkpoint (zeros otherwise -- those are
        read-only stride-0 views without storage: clone before editing in place)."""
        dest = kw.get("destination", args[0] if len(args) > 0 else None)
        if dest is not None:
            # nn.Module.state_dict of a PARENT module (or a caller-supplied destination) ignores what a child returns and keeps `destination`:
            # put the host-side entries into it as well, so nesting dec_ipt does not drop the 208.6 M never-read parameters from a checkpoint
            # (they are appended; the reference's key ORDER is only reproduced by the top-level call below)
            live = super().state_dict(*args, **kw)
            prefix = kw.get("prefix", args[1] if len(args) > 1 else "")
            for key, shape, dt in self._manifest:
                if prefix + key not in live:
                    t = self._dead.get(key)
                    live[prefix + key] = t if t is not None else torch.zeros(shape, dtype=dt)
            return live
        live = super().state_dict(*args, **kw)
        prefix = kw.get("prefix", args[1] if len(args) > 1 else "")
        out = type(live)()
        for key, shape, dt in self._manifest:
            k = prefix + key
            if k in live:
                out[k] = live[k]
            else:
                t = self._dead.get(key)
                out[k] = t if t is not None else torch.zeros((), dtype=dt).expand(shape)       # no storage until someone asks for it
        for k, v in live.items():
            if k not in out:
                out[k] = v
        if hasattr(live, "_metadata"):
            out._metadata = live._metadata
        return out

    def load_state_dict(self, state_dict, strict=True, **kw):
        live, dead = {}, {}
        shapes = {key: shape for key, shape, dt in self._manifest if _is_dead(key)}
        for k, v in state_dict.items():
            if k in shapes:
                if tuple(v.shape) != tuple(shapes[k]):
                    raise RuntimeError("size mismatch for %s: copying a param with shape %s from checkpoint, the shape in current model "
                                       "is %s" % (k, tuple(v.shape), tuple(shapes[k])))
                dead[k] = v.detach().cpu()
            else:
                live[k] = v
        if strict:
            missing = [k for k in shapes if k not in dead]
            if missing:
                raise RuntimeError("Error(s) in loading state_dict for dec_ipt: Missing key(s) in state_dict: %s"
                                   % ", ".join('"%s"' % k for k in missing[:8]) + (" ..." if len(missing) > 8 else ""))
        r = super().load_state_dict(live, strict=strict, **kw)
        self._dead = dead
        self.invalidate()
        return r

    def _apply(self, fn, *a, **kw):
        r = super()._apply(fn, *a, **kw)
        self.invalidate()            # packed copies and nets hold pointers into the old device
        return r

    def invalidate(self):
        """Call after mutating parameters in place; packed copies are rebuilt at the next forward."""
        self._packed = None
        self._packed_cache = {}
        self._free_nets()

    def _free_nets(self):
        lib = _lib.load() if self._nets else None
        for h, _ in self._nets.values():
            lib.cfen_net_destroy(h)
        self._nets = {}
        self._native_u8 = {}
        self._graph_keep = []        # captured graphs died with their nets
        self._graphs = []
        self._last = None

    def __del__(self):
        try:
            self._free_nets()
        except Exception:
            pass

    def set_compute_dtype(self, dtype):
        """Switch the arithmetic type of the following forwards.  Packed weights and launch plans are kept PER TYPE (round 5): the --precision half
        checks of the harness flip between fp16 and fp32 every --half_guard_every batches, and repacking 271 M parameters each time cost seconds."""
        dtype = _DTYPES[dtype]
        if dtype == self.compute_dtype:
            return
        if self._packed is not None and not self._an_pending:
            # (a packed set whose ActNorm layers are still uninitialised is NOT kept: once they are initialised under the other type its epilogue tables and its
            # pending list would be stale -- the next forward of this type would initialise ActNorm again, from another batch; it is repacked instead.  ADVICE r05)
            self._packed_cache[self.compute_dtype] = (self._packed, self._packed_dev, self._an_pending, getattr(self, "_ones", None))
        self.compute_dtype = dtype
        self._packed = None
        hit = self._packed_cache.pop(dtype, None)
        if hit is not None:
            self._packed, self._packed_dev, self._an_pending, self._ones = hit

    def release_other_dtypes(self):
        """free the packed weights, launch plans and workspaces of every compute type but the current one (both types stay resident after a switch: ~0.8 GB of packed
        weights per type plus each replica's workspace -- what the harness's periodic fp32 checks want, not what a one-off fp32 leg needs afterwards)"""
        self._packed_cache = {}
        lib = _lib.load() if self._nets else None
        for key in [k for k in self._nets if k[5] != self.compute_dtype]:
            lib.cfen_net_destroy(self._nets.pop(key)[0])
            self._native_u8.pop(key, None)
        torch.cuda.empty_cache()

    def _live_state(self, device):
        sd = {}
        for k, v in self.state_dict().items():
            if not _is_dead(k):
                sd[k] = v.detach().to(device)
        return sd

    def _ensure_packed(self, device):
        if self._packed is not None and self._packed_dev != device:
            self.invalidate()
        if self._packed is None:
            pending = {}
            packed = pack_state_dict(self._live_state(device), self.cfg, self.compute_dtype, pending=pending, wtile=self.wtile,
                                     gvit_stream=self.gvit_chain)
            self._packed = {k: v.to(device).contiguous() for k, v in packed.items() if not isinstance(v, str)}
            for k, v in packed.items():          # "@other": the same device tensor under a second name (shared modules)
                if isinstance(v, str):
                    self._packed[k] = self._packed[v[1:]]
            self._packed_dev = device
            # uninitialised ActNorm2d layers: (key prefix, conv bias [Cout_pad], an_out [2][Cout_pad]) on the device
            self._an_pending = {n: (an, b.to(device).contiguous(), torch.zeros(2, b.numel(), dtype=torch.float32, device=device))
                                for n, (an, b) in pending.items()}
            self._ones = torch.ones(128, dtype=torch.float32, device=device)
        return self._packed

    def _net_for(self, batch, device, u8=False):
        packed = self._ensure_packed(device)
        key = (batch, bool(u8), bool(self.serial_plan), int(self.replica), bool(self.output_u8), self.compute_dtype, bool(self.output_f16))
        if key in self._nets:
            return self._nets[key]
        if self.gvit_chain:
            # the persistent chains meet at a grid barrier: every team of every forward in flight must be resident at once.  The host caps the team at
            # 256 / (ng x "gvit.max_concurrent") CUs; a replica beyond that number could leave two launches partially resident, spinning on each other
            # until GV_SPIN_LIMIT and then continuing with unsynchronised data (csrc/k_gvit.hip) -- refused here
            from . import ops
            allowed = ops.tuned("gvit.max_concurrent", 1)
            if int(self.replica) >= allowed and ops.tuned("net.gvit_chain", 1) not in (4, 5):      # (4 / 5: one launch per GEMM, no grid barrier)
                raise CfenError("CFEN_GVIT_CHAIN=1: replica %d would put %d chain plans in flight, the grid barriers are sized for %d "
                                "(ops.tune('gvit.max_concurrent', n) BEFORE building the nets caps the teams accordingly)"
                                % (self.replica, int(self.replica) + 1, allowed))
        lib = _lib.load()
        c = self.cfg
        cc = NetConfigC(batch=batch, n_feats=c.n_feats, hidden_dim_ratio=c.hidden_dim_ratio, patch_size=c.patch_size,
                        load_size=c.load_size, num_heads=c.num_heads, dtype=_lib.dtype_code(self.compute_dtype),
                        reserved=(1 if self.serial_plan else 0) | (2 if self.wtile else 0) | (4 if self.gvit_chain else 0) | (_VARIANT_CODE[c.variant] << 8))
        # bit 0: single-stream plan; bit 1: tile-major GViT weights; bit 2: GViT fragment streams (persistent chains); bits 8..15: variant
        h = ctypes.c_void_p()
        check(lib.cfen_net_create(ctypes.byref(h), ctypes.byref(cc)), "cfen_net_create")
        for name, t in packed.items():
            check(lib.cfen_net_set_param(h, name.encode(), ptr(t), t.numel() * t.element_size()), "cfen_net_set_param(%s)" % name)
        buf = ctypes.create_string_buffer(4096)
        if lib.cfen_net_missing_params(h, buf, 4096):
            raise CfenError("packed parameters missing: " + buf.value.decode())
        if u8:
            check(lib.cfen_net_set_input_u8(h, 1), "cfen_net_set_input_u8")
        native_u8 = bool(self.output_u8) and lib.cfen_net_set_output_u8(h, 1) == 0      # refused (fp32 net, small images): fp32 outputs + tensor2im_u8 passes
        if self.output_f16:
            if self.output_u8:
                raise CfenError("output_f16 and output_u8 exclude each other")
            check(lib.cfen_net_set_output_f16(h, 1), "cfen_net_set_output_f16")
        ws = torch.empty(lib.cfen_net_workspace_bytes(h), dtype=torch.uint8, device=device)
        self._nets[key] = (h, ws)
        self._native_u8[key] = native_u8
        return self._nets[key]

    def plan_info(self):
        """the launch plan the NEXT forward / capture of this module uses, and why (the choice must be visible, not ambient)"""
        set_by_caller = self.serial_plan != (bool(os.environ.get("CFEN_SERIAL")) or self._plan_reason.startswith("GPU_MAX_HW_QUEUES"))
        return {"lanes_per_forward": 1 if self.serial_plan else 2, "replica": int(self.replica),
                "why": "set by the caller (bench.py / pipeline.py: several forwards in flight)" if set_by_caller else self._plan_reason,
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "4 (runtime default)"), "gvit_chain": bool(self.gvit_chain),
                "compute_dtype": str(self.compute_dtype).replace("torch.", "")}

    def chain_errors(self):
        """error words of the persistent GViT chains of the net that ran last (csrc/k_gvit.hip): all zero unless a grid-barrier wait gave up
        (synchronises; for tests and bench.py's self-check, not for the hot path)"""
        if self._last is None:
            return [0, 0, 0]
        h, _ = self._nets[self._last]
        words = (ctypes.c_void_p * 3)()
        check(_lib.load().cfen_net_chain_error_words(h, words, 3), "cfen_net_chain_error_words")
        torch.cuda.synchronize()
        ws = self._nets[self._last][1]          # the words live in the net's workspace tensor: read them through a view of it
        return [int(ws[w - ws.data_ptr():w - ws.data_ptr() + 4].view(torch.int32).item()) for w in words]

    def _arm_actnorm_init(self, h):
        """Uninitialised ActNorm2d layers are initialised by the next eager forward, from that batch (models/actnorm.py:25-37)."""
        lib = _lib.load()
        for name, (an, bias, an_out) in self._an_pending.items():
            check(lib.cfen_net_actnorm_pending(h, name.encode(), ptr(self._ones), ptr(bias), ptr(an_out)), "cfen_net_actnorm_pending")

    def _finish_actnorm_init(self):
        """Copy the device-computed ActNorm parameters into the module (so state_dict() / save_networks see what the reference's
        first forward would have left) and flip `initialized`.  The packed epilogue tables were updated in place by the kernel."""
        mods = dict(self.named_modules())
        for name, (an, bias, an_out) in self._an_pending.items():
            m = mods[an]
            c = m.weight.numel()
            with torch.no_grad():
                m.weight.copy_(an_out[0, :c])
                m.bias.copy_(an_out[1, :c])
                m.initialized.fill_(1)
        self._packed_cache = {}      # sets of the other compute types were packed before these parameters existed
        self._an_pending = {}

    # ---- forward ----------------------------------------------------------------------------
    KERNEL_CLASSES = ("gemm", "attention", "layernorm", "tokens", "conv", "norm", "mlp")

    def forward(self, x, out=None):
        """`out`: optional flat float32 CUDA buffer of 7*B*H*W elements that receives [xr | xs | xd]
        back to back (one slab for the data-parallel all-gather); the returned tensors are views of it."""
        return self._run(x, out, None)

    def capture(self, x, out=None):
        """Build + instantiate the launch plan as a hipGraph for these exact tensors (native kernel nodes with
        explicit dependencies, csrc/cfen_net.cpp; not torch's stream capture).  Returns (graph id, [xr, xs, xd]);
        `x` and the outputs must stay alive and in place while the graph is replayed."""
        res = {}
        outs = self._run(x, out, res, capture=True)
        self._graph_keep.append((x, outs))
        self._graphs.append((self._last, res["gid"]))
        return len(self._graphs) - 1, outs

    def replay(self, gid):
        key, native = self._graphs[gid]
        h, _ = self._nets[key]
        check(_lib.load().cfen_net_graph_launch(h, native, current_stream()), "cfen_net_graph_launch")

    def profile(self, x):
        """One forward with HIP events around every launch: {class: (ms, algorithmic flops, launches)}."""
        prof = {}
        self._run(x, None, prof)
        return prof

    def _run(self, x, out, prof, capture=False):
        if not x.is_cuda:
            raise CfenError("the HIP generator needs a CUDA(HIP) tensor; there is no CPU fallback (got %s)" % x.device)
        n = self.cfg.image_size
        u8 = x.dtype == torch.uint8          # (B,H,W,3) uint8 as decoded from the file: normalised on the device (data/base_dataset.py:44-46)
        want = (n, n, self.cfg.n_colors) if u8 else (self.cfg.n_colors, n, n)
        if x.dim() != 4 or tuple(x.shape[1:]) != want:
            raise RuntimeError("input must be (B,%d,%d,%d) float or (B,%d,%d,%d) uint8 for --loadSize %d --patch_size %d (image size is "
                               "baked into the network, reference v3:1186); got %s %s" % (self.cfg.n_colors, n, n, n, n, self.cfg.n_colors,
                                                                                       self.cfg.load_size, self.cfg.patch_size,
                                                                                       tuple(x.shape), x.dtype))
        if capture and (not x.is_contiguous() or x.dtype not in (torch.float32, torch.uint8)):
            raise ValueError("capture() needs a contiguous float32 / uint8 input (its address is baked into the graph)")
        x = x.contiguous() if u8 else x.contiguous().float()
        B = x.shape[0]
        h, ws = self._net_for(B, x.device, u8)
        init_actnorm = bool(self._an_pending)
        if init_actnorm:
            if capture or prof is not None:
                raise CfenError("ActNorm2d layers are uninitialised: run one plain forward (it initialises them from its batch, "
                                "models/actnorm.py:25-37) before capture() / profile()")
            self._arm_actnorm_init(h)
        px = B * n * n
        key = (B, bool(u8), bool(self.serial_plan), int(self.replica), bool(self.output_u8), self.compute_dtype, bool(self.output_f16))
        native_u8 = self._native_u8[key]
        if self.output_u8 and out is not None:
            raise ValueError("output_u8 allocates its own (B,H,W,3) uint8 outputs: no `out` slab")
        if capture and self.output_u8 and not native_u8:
            raise CfenError("capture() with output_u8 needs a net that writes the uint8 images itself (fp16, full-size tails); this one converts its "
                            "float outputs with separate passes AFTER the forward, which a replayed graph would not run -- capture with output_u8 = "
                            "False and convert after the replay (pipeline.py does)")
        if native_u8:
            flat = torch.empty(9 * px, dtype=torch.uint8, device=x.device)
            xr, xs, xd = (flat[k * 3 * px:(k + 1) * 3 * px].view(B, n, n, 3) for k in range(3))
        else:
            odt = torch.float16 if self.output_f16 else torch.float32
            if out is None:
                out = torch.empty(7 * px, dtype=odt, device=x.device)
            elif out.dtype != odt or out.numel() != 7 * px or not out.is_contiguous() or out.device != x.device:
                raise ValueError("out must be a contiguous %s buffer of 7*B*H*W elements on the input's device" % ("float16 (output_f16)" if self.output_f16 else "float32"))
            flat = out.view(-1)
            xr, xs, xd = flat[:3 * px].view(B, 3, n, n), flat[3 * px:4 * px].view(B, 1, n, n), flat[4 * px:].view(B, 3, n, n)
        lib = _lib.load()
        if capture:
            gid = ctypes.c_int32()
            check(lib.cfen_net_graph_capture(h, ptr(x), ptr(xr), ptr(xs), ptr(xd), ptr(ws), ws.numel(), ctypes.byref(gid)),
                  "cfen_net_graph_capture")
            prof["gid"] = gid.value
        elif prof is None:
            check(lib.cfen_net_forward(h, ptr(x), ptr(xr), ptr(xs), ptr(xd), ptr(ws), ws.numel(), current_stream()), "cfen_net_forward")
        else:
            nc = len(self.KERNEL_CLASSES)
            ms, fl, cnt = (ctypes.c_double * nc)(), (ctypes.c_double * nc)(), (ctypes.c_int32 * nc)()
            check(lib.cfen_net_profile(h, ptr(x), ptr(xr), ptr(xs), ptr(xd), ptr(ws), ws.numel(), current_stream(), ms, fl, cnt, nc),
                  "cfen_net_profile")
            for i, name in enumerate(self.KERNEL_CLASSES):
                prof[name] = (ms[i], fl[i], cnt[i])
            detail, i = [], 0
            lab, cls, f, t = c_char_p_(), ctypes.c_int32(), ctypes.c_double(), ctypes.c_double()
            while lib.cfen_net_profile_entry(h, i, ctypes.byref(lab), ctypes.byref(cls), ctypes.byref(f), ctypes.byref(t)) == 0:
                kn, by = c_char_p_(), ctypes.c_double()
                lib.cfen_net_profile_entry_kernel(h, i, ctypes.byref(kn), ctypes.byref(by))
                detail.append((lab.value.decode(), self.KERNEL_CLASSES[cls.value], f.value, t.value, (kn.value or b"?").decode(), by.value))
                i += 1
            prof["launches"] = detail
        if init_actnorm:
            self._finish_actnorm_init()
        self._last = key
        if self.output_u8 and not native_u8:
            from . import ops
            return [torch.stack([ops.tensor2im_u8(t[b].contiguous()) for b in range(B)]) for t in (xr, xs, xd)]
        return [xr, xs, xd]

    def set_scale(self, scale_idx):
        self.scale_idx = scale_idx

    # ---- introspection (parity tests, roofline) -----------------------------------------------
    def stage(self, name):
        """NCHW float32 copy of a top-level stage output of the last forward (SURVEY Appendix D names)."""
        if self._last is None:
            raise CfenError("no forward has run")
        h, ws = self._nets[self._last]
        p = ctypes.c_void_p()
        C, cs, H, W = (ctypes.c_int32() for _ in range(4))
        check(_lib.load().cfen_net_stage(h, name.encode(), ctypes.byref(p), ctypes.byref(C), ctypes.byref(cs), ctypes.byref(H),
                                         ctypes.byref(W)), "cfen_net_stage")
        dt = self._last[5]
        esz = 2 if dt == torch.float16 else 4
        off = p.value - ws.data_ptr()
        B = self._last[0]
        n = B * H.value * W.value * cs.value
        flat = ws[off:off + n * esz].view(dt)
        return flat.view(B, H.value, W.value, cs.value)[..., :C.value].permute(0, 3, 1, 2).float().contiguous()

    def writes_u8_natively(self):
        """did the net of the last forward write the uint8 images itself (tensor2im in the tails' last launch)?"""
        return self._last is not None and bool(self._native_u8.get(self._last, False))

    def flops_per_image(self):
        if not self._nets:
            raise CfenError("no net instantiated yet")
        h, _ = next(iter(self._nets.values()))
        return float(_lib.load().cfen_net_flops_per_image(h))


def init_weights(net, init_type="kaiming", gain=0.02, seed=None):
    """Counterpart of init_weights (v3:49-74) + nn defaults for what it leaves alone.  Conv/Linear
    weights: `init_type`; their biases 0; LayerNorm 1/0; MHA in_proj kaiming_uniform(a=sqrt(5)) (v3:1377);
    embeddings N(0,1).  ActNorm stays uninitialised (`initialized` = 0) exactly like the reference."""
    import math
    g = torch.Generator()
    if seed is not None:
        g.manual_seed(seed)
    with torch.no_grad():
        for k, p in net.named_parameters():
            leaf = k.rsplit(".", 1)[-1]
            if k.startswith(("sub_mean", "add_mean")):
                continue
            if p.dim() >= 2:
                fan_in = p[0].numel()
                if ".pe.weight" in k or "query_embed" in k:
                    p.copy_(torch.randn(p.shape, generator=g))
                elif "in_proj_weight" in k:
                    b = 1.0 / math.sqrt(fan_in)
                    p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * b)
                elif init_type == "kaiming":
                    p.copy_(torch.randn(p.shape, generator=g) * math.sqrt(2.0 / fan_in))
                elif init_type == "normal":
                    p.copy_(torch.randn(p.shape, generator=g) * gain)
                elif init_type == "xavier":
                    fan_out = p.shape[0] * (p[0][0].numel() if p.dim() > 2 else 1)
                    p.copy_(torch.randn(p.shape, generator=g) * gain * math.sqrt(2.0 / (fan_in + fan_out)))
                elif init_type == "orthogonal":
                    nn.init.orthogonal_(p, gain=gain)
                else:
                    raise NotImplementedError("initialization method [%s] is not implemented" % init_type)
            elif ".norm" in k:
                p.fill_(1.0 if leaf == "weight" else 0.0)
            elif leaf == "bias":
                p.zero_()
    print("initialize network with %s" % init_type)
    if hasattr(net, "invalidate"):
        net.invalidate()


def define_G(opt, conv=None, compute_dtype=None):
    """Counterpart of define_G/init_net (v3:93-100, 77-83): build, move to GPU, initialise.  Multi-GPU
    is one process per GPU (parallel.py), not nn.DataParallel, so gpu_ids[0] is the only device used."""
    if compute_dtype is None:
        compute_dtype = {"single": "fp32", "half": "fp16"}.get(getattr(opt, "precision", "single"), "fp32")
    net = dec_ipt(opt, conv, compute_dtype=compute_dtype)
    gpu_ids = getattr(opt, "gpu_ids", [])
    if len(gpu_ids) > 1:
        raise NotImplementedError("--gpu_ids %s: the reference wraps the net in nn.DataParallel (v3:77-83); here multi-GPU is one process "
                                  "per GPU: `python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 test.py "
                                  "--sb ...` (every rank takes the GPU of its LOCAL_RANK and its own slice of the images, "
                                  "cfen_vit_dehazing_amd/parallel.py)" % (gpu_ids, len(gpu_ids)))
    if len(gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.to(gpu_ids[0])
    init_weights(net, getattr(opt, "init_type", "kaiming"))
    return net
