#!/usr/bin/env python3
"""Headline benchmark: images/sec of the CFEN-ViT v3 generator forward at 512x512, n_feats=24
(BASELINE.json configs[1]: batch 8 per GPU, hidden_dim_ratio 4, fp16 storage / fp32 accumulate).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one forward of one batch of synthetic hazy tensors already resident in HBM, through
libcfen_hip.so (replayed from a hipGraph).  Consecutive steps rotate over --in-flight launch plans (own workspace
and output slab, shared weights) on as many streams, so several forwards are in flight at once (default: 4 on one
GPU, 3 per rank beside the all-gather stream; one hardware queue per stream: GPU_MAX_HW_QUEUES=8 is exported below);
all K steps complete inside the timed region, and the same line reports ONE forward at a time as well
(pipelining.one_forward_in_flight).  With N > 1 every rank runs its own 8 images (weak scaling,
weights replicated) and the per-rank output slab is all-gathered with RCCL; the gather of step i is launched from
the lane that ran step i and overlaps the forwards of the other lanes; the last ones are waited for inside the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The pipelined run keeps --in-flight forwards on as many streams; the HIP runtime deals a process's streams onto GPU_MAX_HW_QUEUES hardware
# queues (default 4), and two streams on one queue run one behind the other.  Measured on MI355X (profiles/r04_ab_hw_queues.txt): 4 forwards in
# flight 2.51 ms / step on 4 queues, 2.10 on 8 (3 in flight: 2.18 either way).  Must be in the environment before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

MFMA_PEAK_TFLOPS = {"fp16": 2500.0, "fp32": 157.3}     # /opt/skills/guides/MI355X_MICROARCH.md chip table (dense)


def cpu_baseline(cfg, seconds=20.0):
    """The CPU oracle (a port of the reference forward, pinned to it by tests/golden) timed on the
    host cores on a bounded sample: batch-1 forwards at the benchmark's image size for ~`seconds`.
    torch's default thread count on a 128-core host oversubscribes a batch-1 forward (0.15 img/s there against the reference's
    0.93 img/s on 8 cores): two thread counts are tried and the faster one is the reported value."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import cfen_oracle
    from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
    sd = generate_state_dict(cfg, seed=0, with_dead=False)
    x = synthetic_input(1, cfg)
    ncpu = os.cpu_count() or 8
    tried = {}
    default_threads = torch.get_num_threads()
    try:
        for nt in sorted({min(8, ncpu), min(32, ncpu)}):
            torch.set_num_threads(nt)
            with torch.no_grad():
                cfen_oracle.forward(sd, x, cfg.num_heads, cfg.patch_size)          # warm-up
                times = []
                t_end = time.time() + seconds / 2
                while time.time() < t_end and len(times) < 15:
                    t0 = time.time()
                    cfen_oracle.forward(sd, x, cfg.num_heads, cfg.patch_size)
                    times.append(time.time() - t0)
            times.sort()
            tried[nt] = (1.0 / times[len(times) // 2], len(times))
    finally:
        torch.set_num_threads(default_threads)
    best = max(tried, key=lambda k: tried[k][0])
    return {"value": round(tried[best][0], 4), "unit": "images/sec", "cores": best, "kind": "port",
            "sample": "batch-1 fp32 forwards of the CPU oracle at %dx%d, median per thread count: %s (host has %d logical cores)"
                      % (cfg.image_size, cfg.image_size,
                         ", ".join("%d threads %.3f img/s over %d runs" % (k, v[0], v[1]) for k, v in sorted(tried.items())), ncpu)}


def kernel_key(label, cls):
    """per-launch label of cfen_net_profile ("localvit_decoder_02r (x3):proj_mlp_fused", "tail_R.conv7 (x3)") -> the kernel it ran:
    block kind + level + step for transformer launches, the layer family for convolutions"""
    name, _, step = label.partition(":")
    name = name.split(" ")[0]
    if name.startswith(("localvit", "globalvit")):
        return "%s%s:%s" % ("lvit" if name.startswith("local") else "gvit", name.split("_0")[1][0], step)
    fam = name.rstrip("0123456789rsd") if not name.startswith("tail_") else "tail." + name.split(".")[-1]
    return "%s:%s" % (cls, fam + (":" + step if step else ""))


HBM_PEAK_GBS = 8000.0                                   # same guide: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def nearer_roof(gflop, gbytes, ms, peak_tflops):
    """(bound, achieved, peak, unit, frac) of a set of launches: the roof it is nearer to -- MFMA (algorithmic flops / time against the dense peak of
    the dtype) or HBM (algorithmic bytes / time against 8 TB/s; only launches the net prices in bytes carry them: the weight-streaming token GEMMs)"""
    f_m = gflop / ms / peak_tflops if ms > 0 else 0.0
    f_h = gbytes / ms * 1e3 / HBM_PEAK_GBS if ms > 0 else 0.0          # GB / ms * 1e3 = GB/s
    if f_h > f_m:
        return "hbm", round(gbytes / ms * 1e3, 1), HBM_PEAK_GBS, "GB/s", round(f_h, 5)
    return "mfma", round(gflop / ms, 2), peak_tflops, "TFLOP/s", round(f_m, 5)


def run_child(extra_args, timeout=420):
    """one optional leg of the default run as a FRESH PROCESS of this same file in --brief mode (round 6): the leg gets the streams, hardware queues and
    allocator state of a stand-alone run (inside the headline process configs 4 / 5 measured 6-9 % below their stand-alone runs on the same box), and a hang or an RCCL
    abort in it is a timeout here, not a lost headline line.  Returns the child's JSON (or {"error": ...})."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--brief", "--no-cpu-baseline", "--no-extra-configs"] + [str(a) for a in extra_args]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    except subprocess.TimeoutExpired:
        return {"error": "timed out after %d s: %s" % (timeout, " ".join(cmd[2:]))}
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-400:])}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": "unparsable result line: %s" % e}


def self_check(net, x, outs, cfg, dtype):
    """The benchmarked computation is validated where it is timed: the slab the LAST graph replay wrote against (a) image 0's own batch-1
    eager forward and (b) the reference vectors of tests/golden (made by importing the reference, tools/gen_golden.py): EVERY image of the batch
    where the configuration has a fixture of that batch (the headline: net_full512b8_*, seeds 0 .. 7 = synthetic_input's), image 0 otherwise.
    Returns max-abs differences."""
    import numpy as np
    res = {}
    one = net(x[0:1].clone())
    res["image0_vs_batch1_eager"] = max(float((a[0:1] - b).abs().max()) for a, b in zip(outs, one))
    stem = "%sfull%d" % ("" if cfg.variant == "v3" else cfg.variant + "_", cfg.image_size)
    tail = "_nf%d_hdr%d.npz" % (cfg.n_feats, cfg.hidden_dim_ratio)
    B = int(outs[0].shape[0])
    fix_b = os.path.join(ROOT, "tests", "golden", "net_%sb%d%s" % (stem, B, tail))
    fix_1 = os.path.join(ROOT, "tests", "golden", "net_%s%s" % (stem, tail))
    fix, nimg = (fix_b, B) if os.path.exists(fix_b) else (fix_1, 1)
    if os.path.exists(fix):
        z = np.load(fix)
        n = cfg.image_size
        c0 = n // 2 - 32
        per_image = [0.0] * nimg
        for nm, o in zip(("xr", "xs", "xd"), outs):
            oc = o[0:nimg].float().cpu()
            d2 = (oc[:, :, 3::8, 5::8] - torch.from_numpy(z["strided/" + nm])).abs().flatten(1).max(1).values
            d1 = (oc[:, :, c0:c0 + 64, c0:c0 + 64] - torch.from_numpy(z["crop/" + nm])).abs().flatten(1).max(1).values if ("crop/" + nm) in z else d2
            per_image = [max(p, float(a), float(b)) for p, a, b in zip(per_image, d1, d2)]
        res["image0_vs_reference_vectors"] = per_image[0]
        if nimg > 1:
            res["images_vs_reference_vectors"] = [round(v, 7) for v in per_image]
            res["images_checked_against_reference"] = nimg
    bar = 5e-3 if dtype == "fp16" else 1e-3
    res["bar"] = bar
    res["ok"] = all(v <= bar for k, v in res.items() if k.startswith("image0")) and all(v <= bar for v in res.get("images_vs_reference_vectors", []))
    if getattr(net, "gvit_chain", False):       # the persistent-chain variant: a grid-barrier wait that gave up leaves a mark (csrc/k_gvit.hip)
        res["chain_error_words"] = net.chain_errors()
        res["ok"] = res["ok"] and not any(res["chain_error_words"])
    return res


def pmc_traffic(kernel):
    """HBM bytes per launch of a device kernel from the newest committed rocprofv3 PMC summary (tools/mfma_summary.py: separate FETCH_SIZE and
    WRITE_SIZE passes of this same command, fetch doubled as the gfx950 guide prescribes).  `kernel` is the launch-site name cfen_net_profile
    records ("k_gemm_dma<T, 1, 3>"); the summary is keyed by the profiler's symbol ("k_gemm_dma<f16,1,3,3>": element type spelled out,
    defaulted template arguments included) -- matched on the kernel name and the leading numeric arguments.  A cross-reference to profiles/,
    not a live measurement: None without a match."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None

    def parts(name):
        base, _, args = name.partition("<")
        nums = [a.strip() for a in args.rstrip(">").split(",") if re.fullmatch(r"\s*-?\d+\s*", a)]
        return base.strip(), nums
    try:
        table = json.load(open(files[-1]))["kernels"]
    except (KeyError, ValueError, OSError):
        return None
    base, nums = parts(kernel.split(" + ")[0])
    for key, val in table.items():
        kb, kn = parts(key)
        if kb == base and kn[:len(nums)] == nums:
            return val.get("hbm_bytes_per_launch")
    return None


def apply_tuning():
    """CFEN_TUNE="key=value,..." -> cfen_tune (kernel-variant experiments; unset = shipped defaults)"""
    spec = os.environ.get("CFEN_TUNE", "")
    if spec:
        from cfen_vit_dehazing_amd import ops
        for kv in spec.split(","):
            k, v = kv.split("=")
            ops.tune(k.strip(), int(v))


def main():
    # stdout carries exactly ONE line, the result: whatever libraries print there while the run lasts (RCCL's version banner when a communicator comes up)
    # is sent to stderr; the JSON line goes to the saved descriptor at the end
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the K-step timed region is repeated until this much time has been measured; value = median repetition")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8, help="images per GPU")
    ap.add_argument("--dtype", default="fp16", choices=["fp16", "fp32"])
    ap.add_argument("--hidden-dim-ratio", type=int, default=4)
    ap.add_argument("--load-size", type=int, default=256, help="256 -> 512x512 images")
    ap.add_argument("--variant", default="v3", choices=["v3", "cfs", "crs", "v5"],
                    help="generator: v3 = iid_hlgvit_crs_gd4_cfs_v3 (BASELINE's); the siblings are informational runs (cfs / crs: the image edge is --load-size)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying a hipGraph")
    ap.add_argument("--gather-dtype", default="auto", choices=["auto", "fp16", "fp32"],
                    help="wire type of the output all-gather (N > 1); auto = the compute dtype")
    ap.add_argument("--in-flight", type=int, default=0, choices=[0, 1, 2, 3, 4, 5, 6, 7, 8],
                    help="forwards in flight (0 = default: 4 on one GPU, 3 per rank when an all-gather stream runs beside them): N > 1 = consecutive steps rotate over N launch plans (own workspace and output slab each, shared "
                         "weights) on N streams, so the tail of step i overlaps the head of steps i + 1 .. i + N - 1; every step is still one whole forward of "
                         "one batch and all K steps complete inside the timed region")
    ap.add_argument("--lanes", type=int, default=0, choices=[0, 1, 2],
                    help="launch plan of one forward: 2 = GViT beside LViT on a second graph branch, 1 = one serial chain of launches; 0 (default) = 1 when "
                         "several forwards are in flight (measured: 2.26 against 2.38 ms per step -- the forwards overlap each other, and a branch in "
                         "every graph costs ~0.05 ms of cross-queue synchronisation per level), 2 for --in-flight 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short timed legs of BASELINE configs 4 (batch 4, 1024x1024) and 5 (batch 16, hidden_dim_ratio 2) in the default run")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--brief", action="store_true",
                    help="a leg of another run: the timed region and the self-check only (no per-launch profile, no one-forward leg), a compact result line")
    ap.add_argument("--force-gather", action="store_true",
                    help="with --gpus 1: run the sharded run's per-rank configuration on ONE GPU -- a world-1 RCCL communicator and the output gatherer behind every step")
    args = ap.parse_args()
    apply_tuning()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if args.in_flight == 0:
        # one hardware queue per concurrently busy stream is what the card runs well: 4 forward lanes on one GPU, 3 beside the communication stream of
        # the sharded run (measured with a stand-in for it, CFEN_BENCH_FAKE_COMM_CYCLES: 4 lanes + 1 busy queue 2.52 ms / step, 3 + 1 2.25; profiles/r04_ab_hw_queues.txt)
        args.in_flight = 4 if world == 1 else 3
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    forced_pg = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    pg_late = bool(os.environ.get("CFEN_BENCH_PG_LATE"))     # what-if probe: the communicator (and the stream torch gives it) created AFTER the lanes and their graphs

    def init_forced_pg():
        import tempfile
        import torch.distributed as fpg                  # a file store: no port to race for
        store_dir = tempfile.mkdtemp(prefix="cfen_bench_store_")
        fpg.init_process_group("nccl", store=fpg.FileStore(os.path.join(store_dir, "store"), 1), rank=0, world_size=1, device_id=dev)
        return fpg
    if world == 1 and args.force_gather and not pg_late:
        forced_pg = init_forced_pg()

    from cfen_vit_dehazing_amd.config import NetConfig
    from cfen_vit_dehazing_amd.hipnet import dec_ipt
    from cfen_vit_dehazing_amd.manifest import generate_state_dict, synthetic_input
    from cfen_vit_dehazing_amd.parallel import OutputGatherer

    cfg = NetConfig(24, args.hidden_dim_ratio, patch_size=args.load_size // 8, load_size=args.load_size, variant=args.variant)
    B, n = args.batch, cfg.image_size
    net = dec_ipt(cfg, compute_dtype=args.dtype)
    net.load_state_dict(generate_state_dict(cfg, seed=0), strict=True)
    net.to(dev)
    lanes_plan = args.lanes or (1 if args.in_flight > 1 else 2)
    if not os.environ.get("CFEN_SERIAL"):
        net.serial_plan = lanes_plan == 1
    x = synthetic_input(B, cfg, seed0=rank * B).to(dev)
    nslab = max(2, args.in_flight)
    gdt = args.dtype if args.gather_dtype == "auto" else args.gather_dtype
    # the sharded run gathers fp16: the fused tail launch writes that type itself (round 6: no conversion pass, no fp32 slab) where the geometry allows
    wire_f16 = (world > 1 or args.force_gather) and gdt == "fp16" and args.dtype == "fp16" and n % 64 == 0 and args.variant == "v3" and not os.environ.get("CFEN_BENCH_FP32_SLABS")
    net.output_f16 = wire_f16
    slabs = [torch.empty(7 * B * n * n, dtype=torch.float16 if wire_f16 else torch.float32, device=dev) for _ in range(nslab)]
    def make_gather():
        return OutputGatherer(world, slabs[0].numel(), dev, torch.float16 if gdt == "fp16" else torch.float32, slots=nslab)
    gather = make_gather() if (world > 1 or (args.force_gather and not pg_late)) else None

    net(x, out=slabs[0])                       # packs weights, builds the plan
    torch.cuda.synchronize()
    # what-if probe (CFEN_BENCH_DUMMY_STREAMS=n, not a benchmark setting): n extra streams are created and touched before the graphs are built, so that the
    # hardware queues the runtime hands to the two-lane graph's branch streams are already shared -- the difference between `--in-flight 1` on 8 queues
    # (4.49 ms) and the one-forward leg of the default run, which holds four lane streams (2.75 ms): profiles/r05_ab_hw_queue_placement.txt
    dummies = [torch.cuda.Stream(dev) for _ in range(int(os.environ.get("CFEN_BENCH_DUMMY_STREAMS", "0")))]
    for ds in dummies:
        with torch.cuda.stream(ds):
            torch.zeros(16, device=dev).add_(1)
    torch.cuda.synchronize()
    graphs = None
    nfl = args.in_flight
    # CFEN_BENCH_LANE_PRIORITIES="-1,0,0,0" (what-if probe): stream priorities of the lanes (lower = more urgent); default: all equal
    prios = [int(v) for v in os.environ.get("CFEN_BENCH_LANE_PRIORITIES", "").split(",") if v.strip()]
    lanes = [torch.cuda.Stream(dev, priority=prios[k % len(prios)]) if prios else torch.cuda.Stream(dev) for k in range(nfl)] if nfl > 1 else None
    # CFEN_BENCH_CU_MASK="xcd" | "spread" (what-if probe, round 6): every lane gets its own share of the CUs (hipExtStreamCreateWithCUMask; a serial hipGraph replays on its
    # stream's hardware queue) -- "xcd": whole XCDs per lane (mask bit i = CU i / 8 of XCD i % 8), "spread": 32 / nfl CUs of every XCD per lane
    cu_mask_mode = os.environ.get("CFEN_BENCH_CU_MASK", "")
    if cu_mask_mode and nfl > 1:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        ncu, nx = torch.cuda.get_device_properties(dev).multi_processor_count, 8
        parts = int(os.environ.get("CFEN_BENCH_CU_PARTS", str(nfl)))       # lanes k and k + parts share a partition
        lanes = []
        for k in range(nfl):
            bits = [0] * ((ncu + 31) // 32)
            for i in range(ncu):
                xcd, cu = i % nx, i // nx
                mine = (xcd * parts // nx == k % parts) if cu_mask_mode == "xcd" else (cu * parts // (ncu // nx) == k % parts)
                if mine:
                    bits[i // 32] |= 1 << (i % 32)
            arr = (ctypes.c_uint32 * len(bits))(*bits)
            h = ctypes.c_void_p()
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(len(bits)), arr)
            if rc != 0:
                raise RuntimeError("hipExtStreamCreateWithCUMask failed: %d" % rc)
            lanes.append(torch.cuda.ExternalStream(h.value, device=dev))
        sys.stderr.write("lanes on CU masks (%s): %d partitions of %d CUs\n" % (cu_mask_mode, parts, ncu // parts))
    if not args.no_graph:
        graphs = []
        for k, s in enumerate(slabs):          # native hipGraph per output slab (and, with two forwards in flight, per launch-plan replica)
            net.replica = k % nfl
            graphs.append(net.capture(x, out=s)[0])
        net.replica = 0
        torch.cuda.synchronize()

    if world == 1 and args.force_gather and pg_late:
        forced_pg = init_forced_pg()
        gather = make_gather()
    fake_comm = [int(os.environ.get("CFEN_BENCH_FAKE_COMM_CYCLES", "0")), None]
    if fake_comm[0]:
        fake_comm[1] = torch.cuda.Stream(dev)
    use_lanes = [lanes is not None]     # the serial leg of the default run flips this: the same graphs, one stream, one forward at a time

    def step(i):
        k = i % nslab
        s = slabs[k]
        ctx = torch.cuda.stream(lanes[k % nfl]) if (lanes is not None and use_lanes[0]) else None
        if ctx is not None:
            ctx.__enter__()
        try:
            if gather is not None:
                gather.before_write(k)             # slot = slab: the gather of step i - nslab, the last reader of THIS slab, must be done with it
            if graphs is not None:
                net.replay(graphs[k])
            else:
                net.replica = k % nfl
                net(x, out=s)
                net.replica = 0
            if gather is not None:
                gather.launch(s, k)                # wire conversion on this lane, then the all-gather handed to torch asynchronously (parallel.OutputGatherer)
            elif fake_comm[0]:
                # what-if probe (CFEN_BENCH_FAKE_COMM_CYCLES=n, results are not a benchmark line): a one-workgroup kernel of n cycles per step on its own
                # stream behind the forward -- the footprint of an always-busy communication queue beside the --in-flight lanes on ONE GPU
                fake_comm[1].wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(fake_comm[1]):
                    torch.cuda._sleep(fake_comm[0])
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    lane_search = None
    if (gather is not None or os.environ.get("CFEN_BENCH_LANE_SEARCH")) and lanes is not None and graphs is not None and not os.environ.get("CFEN_BENCH_NO_LANE_SEARCH"):
        # WHICH streams the lanes are matters once torch's collective stream is busy beside them: the HIP runtime deals a process's streams onto the hardware queues in an
        # order the caller does not control, and a lane that shares a queue with the collective stream runs one behind the other (measured on one MI355X, world-1 RCCL,
        # 3 lanes: 2.59 ms per step as created, 4.04 / 2.70 / 2.73 / 2.07 with 1 / 2 / 3 / 5 unused streams created first; 2.08 without the gather:
        # profiles/r06_gather_lane_placement.txt).  A hipGraph replays on any stream, so the harness TRIES the placements: candidate streams c0 .. c(nfl+12), the lanes are
        # c[s : s + nfl] for the shift s with the shortest 12-step region (each rank decides for itself; every rank runs the same number of steps and collectives).
        cands = list(lanes) + [torch.cuda.Stream(dev) for _ in range(int(os.environ.get('CFEN_BENCH_LANE_CANDIDATES', '13')))]
        for cs in cands:
            with torch.cuda.stream(cs):
                torch.zeros(16, device=dev).add_(1)
        torch.cuda.synchronize()
        trial = []
        for sft in range(len(cands) - nfl + 1):
            lanes[:] = cands[sft:sft + nfl]
            for i in range(nslab):
                step(i)
            if gather is not None:
                gather.wait_all()
            barrier()
            t0 = time.perf_counter()
            for i in range(12):
                step(i)
            if gather is not None:
                gather.wait_all()
            torch.cuda.synchronize()
            trial.append(round((time.perf_counter() - t0) / 12 * 1e3, 3))
            barrier()
        best = min(range(len(trial)), key=lambda k: trial[k])
        lanes[:] = cands[best:best + nfl]
        lane_search = {"ms_per_step_by_shift": trial, "chosen_shift": best}

    for i in range(args.warmup):
        step(i)
    if gather is not None:
        gather.wait_all()

    def timed_region():
        """EXACTLY args.steps steps between two (barrier + synchronize) brackets; MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        if gather is not None:
            gather.wait_all()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # a 20-step region is 70 ms: clock ramp and launch jitter dominate it.  The region is repeated (every repetition is the
    # contract's K-step measurement) until >= --min-seconds have been timed; the reported value is the MEDIAN repetition.
    reps = [timed_region()]
    while sum(reps) < args.min_seconds and len(reps) < 200:
        if dist is not None:          # all ranks must agree on the number of repetitions
            flag = torch.tensor([1.0 if sum(reps) < args.min_seconds else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            if flag.item() == 0:
                break
        reps.append(timed_region())
    srt = sorted(reps)
    dt = srt[len(srt) // 2]
    ms_step = dt / args.steps * 1e3
    ips = world * B * args.steps / dt

    serial = None
    if nfl > 1 and graphs is not None and not args.brief:
        # ONE forward in flight, on the launch plan that is best for it (GViT beside LViT on a second graph branch): the same K-step region on the
        # default stream, >= 0.4 s, median -- reported beside the headline so that what the overlap of consecutive steps buys is visible in every run
        was_serial = net.serial_plan
        net.serial_plan, net.replica = False, 0
        g1 = [net.capture(x, out=slabs[k])[0] for k in range(2)]
        net.serial_plan = was_serial
        torch.cuda.synchronize()
        main_graphs, main_nslab = graphs, nslab
        graphs, nslab, use_lanes[0] = g1, 2, False
        for i in range(3):
            step(i)
        sreps = [timed_region()]
        while sum(sreps) < min(0.4, args.min_seconds) and len(sreps) < 50:
            if dist is not None:
                flag = torch.tensor([1.0 if sum(sreps) < min(0.4, args.min_seconds) else 0.0], device=dev)
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() == 0:
                    break
            sreps.append(timed_region())
        graphs, nslab, use_lanes[0] = main_graphs, main_nslab, True
        sdt = sorted(sreps)[len(sreps) // 2]
        serial = {"value": round(world * B * args.steps / sdt, 2), "unit": "images/sec", "ms_per_step": round(sdt / args.steps * 1e3, 3),
                  "repetitions": len(sreps), "lanes_per_forward": 2}
        # the headline state again: the last step of the main plan wrote slabs[(steps - 1) % nslab]; the serial leg overwrote slabs 0 and 1 with the
        # same values (same input, same weights), so the self-check below reads what a main-plan step wrote either way -- rerun one to be exact
        for i in range(args.steps - nslab, args.steps):
            step(i)
        if gather is not None:
            gather.wait_all()
        torch.cuda.synchronize()

    result = None
    if rank == 0 and args.brief:
        from cfen_vit_dehazing_amd.parallel import split_slab
        flops_img = net.flops_per_image()
        torch.cuda.synchronize()
        result = {"workload": "batch=%d %dx%d n_feats=24 hidden_dim_ratio=%d %s" % (B, n, n, args.hidden_dim_ratio, args.dtype),
                  "value": round(ips, 2), "unit": "images/sec", "ms_per_step": round(ms_step, 3), "steps": args.steps, "repetitions": len(reps),
                  "timing_method": "median of the repetitions; a fresh process of bench.py --brief", "forwards_in_flight": nfl, "timed_seconds": round(sum(reps), 3),
                  "gflop_per_image": round(flops_img / 1e9, 2), "whole_forward_tflops": round(ips * flops_img / 1e12, 2),
                  "whole_forward_frac": round(ips * flops_img / 1e12 / MFMA_PEAK_TFLOPS[args.dtype], 5)}
        if lane_search is not None:
            result["lane_search"] = lane_search
        if gather is not None:
            last = (args.steps - 1) % nslab
            result["gathered_equals_slab"] = bool(torch.equal(gather.bufs[last].reshape(-1)[:slabs[last].numel()], slabs[last].to(gather.bufs[last].dtype)))
        result["self_check"] = self_check(net, x, split_slab(slabs[(args.steps - 1) % nslab], B, n), cfg, args.dtype)
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    elif rank == 0:
        flops_img = net.flops_per_image()
        # per-kernel-class timing with HIP events on the launch stream (3 profiled forwards, median of sums)
        profs = [net.profile(x) for _ in range(3)]
        classes = {}
        for name in net.KERNEL_CLASSES:
            runs = sorted(p[name][0] for p in profs)
            classes[name] = {"ms": round(runs[1], 4), "launches": profs[0][name][2], "gflop": round(profs[0][name][1] / 1e9, 2)}
        peak = MFMA_PEAK_TFLOPS[args.dtype]
        for c in classes.values():
            c["tflops"] = round(c["gflop"] / c["ms"], 2) if c["ms"] > 0 and c["gflop"] > 0 else None
        # per-KERNEL table from the per-launch records of the median profile: the roofline line names the dominant kernel
        kern = {}
        mid = sorted(profs, key=lambda p: sum(p[n][0] for n in net.KERNEL_CLASSES))[1]
        for label, cls, fl, ms, _kn, _by in mid["launches"]:
            k = kern.setdefault(kernel_key(label, cls), {"ms": 0.0, "gflop": 0.0, "launches": 0})
            k["ms"] += ms; k["gflop"] += fl / 1e9; k["launches"] += 1
        for k in kern.values():
            k["ms"] = round(k["ms"], 4); k["gflop"] = round(k["gflop"], 2)
            k["tflops"] = round(k["gflop"] / k["ms"], 1) if k["gflop"] > 0 and k["ms"] > 0 else None
        # the same launches grouped by the DEVICE KERNEL that ran them (the template instantiation cfen_net_profile recorded at the launch site:
        # every k_gemm_dma<T, 1, 3> launch of the forward is ONE entry, whatever block, level or step it served)
        sym = {}
        for label, cls, fl, ms, kname, nbytes in mid["launches"]:
            k = sym.setdefault(kname, {"ms": 0.0, "gflop": 0.0, "gbyte": 0.0, "launches": 0})
            k["ms"] += ms; k["gflop"] += fl / 1e9; k["gbyte"] += nbytes / 1e9; k["launches"] += 1
        for k in sym.values():
            k["ms"] = round(k["ms"], 4); k["gflop"] = round(k["gflop"], 2); k["gbyte"] = round(k["gbyte"], 4)
            k["bound"], k["achieved"], _, k["unit"], k["frac"] = nearer_roof(k["gflop"], k["gbyte"], k["ms"], peak)
        dom = max(sym, key=lambda k: sym[k]["ms"])                   # the device kernel the forward spends most time in
        d = sym[dom]
        bound, ach, rpeak, runit, rfrac = nearer_roof(d["gflop"], d["gbyte"], d["ms"], peak)
        dom_launch_ms = d["ms"] / d["launches"]
        whole = ips / world * flops_img / 1e12
        from cfen_vit_dehazing_amd.parallel import split_slab
        check = self_check(net, x, split_slab(slabs[(args.steps - 1) % nslab], B, n), cfg, args.dtype)   # what the last timed step wrote
        if gather is not None:
            # the GATHERED buffer of the last timed step, rank by rank: segment r must be rank r's images (synthetic_input seeds r * B ...), so its
            # first image is checked against a batch-1 eager forward of that image computed here -- an ordering bug of the all-gather /
            # merge_gathered on real RCCL shows up in the scaling run itself (reference analogue: nn.DataParallel's gather, v3:77-83)
            from cfen_vit_dehazing_amd.parallel import merge_gathered
            torch.cuda.synchronize()
            merged = merge_gathered(gather.bufs[(args.steps - 1) % nslab].float(), world, B, n)
            per_rank = []
            for r in range(world):
                xi = synthetic_input(1, cfg, seed0=r * B).to(dev)
                one = net(xi)
                per_rank.append(round(max(float((m[r * B:r * B + 1] - o).abs().max()) for m, o in zip(merged, one)), 6))
            check["gathered_image0_of_rank_vs_batch1_eager"] = per_rank
            check["ok"] = check["ok"] and all(v <= check["bar"] for v in per_rank)
        result = {
            "metric": "images/sec @512x512 n_feats=24", "value": round(ips, 2), "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f16" if args.dtype == "fp16" else "f32", "data": "synthetic",
            "config": {"workload": "batch=%d/GPU %dx%d n_feats=24 hidden_dim_ratio=%d %s%s, weights random-init (seeded generator)"
                                   % (B, n, n, args.hidden_dim_ratio, args.dtype, "" if args.variant == "v3" else " generator variant " + args.variant),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "graph": graphs is not None, "forwards_in_flight": nfl, "lanes_per_forward": 1 if net.serial_plan else 2, "hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                       "gather_dtype": (gdt if world > 1 else None), "lane_search": lane_search,
                       "gflop_per_image": round(flops_img / 1e9, 2)},
            "roofline": {"bound": bound, "kernel": dom, "achieved": ach, "peak": rpeak, "unit": runit, "frac": rfrac,
                         "traffic": None,
                         "traffic_from_profiles": pmc_traffic(dom),
                         "traffic_source": "`traffic` is not measured in this run (null); traffic_from_profiles = HBM bytes per launch of this kernel in the "
                                           "newest committed profiles/r*_pmc_traffic.json (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                           "command, fetch x2 per the gfx950 guide)",
                         "launch_ms": round(dom_launch_ms, 4), "launches_per_step": d["launches"],
                         "algorithmic_gflop_per_launch": round(d["gflop"] / d["launches"], 2),
                         "algorithmic_gbyte_per_launch": round(d["gbyte"] / d["launches"], 4),
                         "mfma_frac": round(d["gflop"] / d["ms"] / peak, 5),
                         "how": "the device kernel with the largest summed launch time in a profiled forward (HIP events on the launch stream around "
                                "every launch, median of three forwards); achieved = its algorithmic flops or bytes / that time, whichever roof is nearer",
                         "whole_forward_tflops": round(whole, 2), "whole_forward_frac": round(whole / peak, 5)},
            "timing": {"method": "value = the MEDIAN of the repetitions of the K-step timed region", "repetitions": len(reps), "timed_seconds": round(sum(reps), 3), "ms_per_step_min": round(srt[0] / args.steps * 1e3, 3),
                       "ms_per_step_max": round(srt[-1] / args.steps * 1e3, 3)},
            "self_check": check,
            "pipelining": {"forwards_in_flight": nfl,
                           "what": "consecutive steps rotate over %d launch plans (own workspace and output slab each, shared weights) on %d streams: the tail "
                                   "of step i overlaps the head of the next ones; every step is one whole batch-%d forward, all K steps finish inside "
                                   "the timed region" % (nfl, nfl, B) if nfl > 1 else "one forward at a time",
                           "one_forward_in_flight": serial},
            "kernel_classes": classes,
            "kernels": dict(sorted(kern.items(), key=lambda kv: -kv[1]["ms"])[:12]),
            # the same launches grouped by the device kernel that ran them (per block shape = per template instantiation)
            "by_symbol": dict(sorted(sym.items(), key=lambda kv: -kv[1]["ms"])[:10]),
        }
        default_run = (B, args.hidden_dim_ratio, args.load_size, args.variant, args.dtype) == (8, 4, 256, "v3", "fp16")
        if world == 1 and default_run and not args.no_extra_configs:
            # the optional legs, each a fresh process of this file (run_child): a failing or hanging leg cannot cost the headline line, and a leg runs with the
            # stream / hardware-queue state of a stand-alone run.  This process keeps its weights and workspaces meanwhile (20 GB of 288).
            extra = {}
            # what the sharded run's communication queue costs a rank, measured instead of modelled: the N = 1 code path WITH the gatherer (a world-1 RCCL
            # communicator, 3 forward lanes, the wire conversion and an asynchronous all_gather_into_tensor behind every step: exactly the per-rank configuration
            # of --gpus N) against the same 3 lanes without it, same box, back to back
            if graphs is not None and nfl >= 3:
                common = ["--steps", args.steps, "--warmup", args.warmup, "--min-seconds", min(0.4, args.min_seconds), "--in-flight", 3]
                legs = {"three_lanes_no_gather": run_child(common), "three_lanes_with_gather": run_child(common + ["--force-gather"])}
                if all("error" not in v for v in legs.values()):
                    legs["ms_per_step_added_by_the_gather_queue"] = round(legs["three_lanes_with_gather"]["ms_per_step"] - legs["three_lanes_no_gather"]["ms_per_step"], 3)
                    legs["gathered_equals_slab"] = legs["three_lanes_with_gather"].get("gathered_equals_slab")
                legs["what"] = ("per-rank configuration of the sharded run on ONE GPU: 3 forwards in flight, each followed on its own lane by the wire conversion and an "
                                "asynchronous RCCL all_gather_into_tensor of the %s output slab (world-1 communicator, torch's internal collective stream is the "
                                "fourth busy queue), one gather slot per slab" % gdt)
                extra["gather_overhead_1gpu"] = legs
            # BASELINE configs 4 and 5, and the exact-fp32 path (the one that meets north_star's 1e-3 max-abs criterion) at the headline batch, so that their
            # numbers are observed by whoever runs bench.py.  40 / 20-step regions: with 4 forwards in flight the fill and drain of a region is about one forward
            for key, eargs in (("config4_batch4_1024x1024", ["--batch", 4, "--load-size", 512, "--steps", 20]),
                               ("config5_batch16_hdr2", ["--batch", 16, "--hidden-dim-ratio", 2, "--steps", 40]),
                               ("fp32_batch8", ["--dtype", "fp32", "--steps", 8])):
                extra[key] = run_child(eargs + ["--warmup", 3, "--min-seconds", 0.4, "--in-flight", nfl])
            result["extra_configs"] = extra
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(cfg, args.cpu_seconds)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(result) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if forced_pg is not None:
        if gather is not None:
            gather.wait_all()
        torch.cuda.synchronize()
        forced_pg.destroy_process_group()


if __name__ == "__main__":
    main()
