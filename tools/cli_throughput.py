#!/usr/bin/env python3
"""File -> file throughput of the inference CLI (test.py), sequential loop against the pipelined driver (--in_flight 4), with the component
rates that bound it measured on the same box:

    python3 tools/cli_throughput.py [n_images=256] [out.json]

Writes synthetic 512 x 512 hazy PNGs + a seeded checkpoint into a scratch directory, then times
  decode_only      the DataLoader alone (PNG decode in --nThreads workers), images/s
  encode_only      PNG encode + write of 512 x 512 RGB images in W writer threads, images/s
  sequential       python test.py --batchSize 8 (one batch at a time, encode on the main thread): the reference's loop; also with --nThreads T (decode in workers)
  pipelined        python test.py --batchSize 8 --in_flight 4 --nThreads T --writers W
  pipelined_procs  ... --nThreads T2 --writer_procs P (round 6: PNG encode in processes forked before the model exists, images through a shared-memory ring)
both CLI runs with --precision half --u8_input --out_all (uint8 in, tensor2im bytes out of the tails' last launch), and checks that the two
result directories hold byte-identical files.  The GPU-only rate of the same batches is bench.py's headline; this tool says how much of it a
file-to-file run sees and which stage is the bound.  Prints / writes ONE JSON object."""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image

from cfen_vit_dehazing_amd.config import NetConfig
from cfen_vit_dehazing_amd.manifest import generate_state_dict

def main():
    n_images = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    ncpu = os.cpu_count() or 8
    threads = int(os.environ.get("CLI_THREADS", min(16, max(2, ncpu // 2))))
    writers = int(os.environ.get("CLI_WRITERS", min(32, max(4, ncpu // 2))))
    tmp = tempfile.mkdtemp(prefix="cfen_cli_")
    try:
        cfg = NetConfig(24, 4, patch_size=32, load_size=256)
        name = "iid_hlgvit_crs_gd4_cfs_v3_cli"
        os.makedirs(os.path.join(tmp, "ckpt", name))
        torch.save(generate_state_dict(cfg, seed=0), os.path.join(tmp, "ckpt", name, "32_net_G.pth"))
        hazy = os.path.join(tmp, "data", "hazy")
        os.makedirs(hazy)
        rs = np.random.RandomState(1)
        n = 512
        yy, xx = np.mgrid[0:n, 0:n].astype(np.float32) / n
        for i in range(n_images):
            base = np.stack([np.sin(6.0 * (xx * (i % 5 + 1) + yy)) * 90 + 128, np.cos(5.0 * (yy * (i % 3 + 1) - xx)) * 80 + 120, (xx + yy) * 100 + 20 + (i % 50)], -1)
            Image.fromarray(np.clip(base + rs.randn(n, n, 3) * 6.0, 0, 255).astype(np.uint8)).save(os.path.join(hazy, "syn_%05d.png" % i))
        res = {"images": n_images, "image": "512x512", "host_logical_cores": ncpu, "decode_workers": threads, "writer_threads": writers}

        # ---- component rates --------------------------------------------------------------------------------------------------------------------
        from cfen_vit_dehazing_amd.options.test_options import TestOptions
        from cfen_vit_dehazing_amd import data as cdata
        common = ["--dataroot", os.path.join(tmp, "data"), "--name", name, "--n_feats", "24", "--hidden_dim_ratio", "4", "--sb", "--out_all", "--which_epoch", "32",
                  "--checkpoints_dir", os.path.join(tmp, "ckpt"), "--precision", "half", "--u8_input", "--batchSize", "8"]
        opt = TestOptions().parse(common + ["--nThreads", str(threads), "--gpu_ids", "-1"])
        t0 = time.perf_counter()
        cnt = sum(len(b["B_paths"]) for b in cdata.CreateDataLoader(opt).load_data())
        res["decode_only_images_per_s"] = round(cnt / (time.perf_counter() - t0), 1)
        from concurrent.futures import ThreadPoolExecutor
        imgs = [np.asarray(Image.open(os.path.join(hazy, "syn_%05d.png" % i)).convert("RGB")) for i in range(min(64, n_images))]
        os.makedirs(os.path.join(tmp, "enc"))
        with ThreadPoolExecutor(writers) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda k: Image.fromarray(imgs[k % len(imgs)]).save(os.path.join(tmp, "enc", "e_%05d.png" % k)), range(n_images)))
            res["encode_only_images_per_s"] = round(n_images / (time.perf_counter() - t0), 1)
        t0 = time.perf_counter()
        for k in range(min(32, n_images)):
            Image.fromarray(imgs[k % len(imgs)]).save(os.path.join(tmp, "enc", "s_%05d.png" % k))
        res["encode_one_thread_images_per_s"] = round(min(32, n_images) / (time.perf_counter() - t0), 1)

        # ---- the two CLI runs -------------------------------------------------------------------------------------------------------------------
        env = dict(os.environ)
        env.pop("GPU_MAX_HW_QUEUES", None)
        procs = int(os.environ.get("CLI_WRITER_PROCS", min(64, max(4, ncpu // 4))))
        threads2 = int(os.environ.get("CLI_THREADS2", min(48, max(2, ncpu // 4))))
        res["writer_processes"], res["decode_workers_with_writer_processes"] = procs, threads2
        for tag, extra in (("sequential", ["--nThreads", "0"]), ("sequential_decode_in_workers", ["--nThreads", str(threads)]),
                           ("pipelined", ["--in_flight", "4", "--nThreads", str(threads), "--writers", str(writers)]),
                           ("pipelined_procs", ["--in_flight", "4", "--nThreads", str(threads2), "--writer_procs", str(procs)]),
                           ("pipelined_procs_png_level_1", ["--in_flight", "4", "--nThreads", str(threads2), "--writer_procs", str(procs), "--png_compress_level", "1"])):
            t0 = time.perf_counter()
            p = subprocess.run([sys.executable, os.path.join(ROOT, "test.py")] + common + ["--results_dir", os.path.join(tmp, "res_" + tag)] + extra,
                               cwd=tmp, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            wall = time.perf_counter() - t0
            if p.returncode != 0:
                res[tag] = {"error": p.stdout[-1500:]}
                continue
            m = re.search(r"(\d+) images in ([0-9.]+) s = ([0-9.]+) images/s file to file", p.stdout)
            mt = re.search(r"main-thread seconds (\{.*\})", p.stdout)
            su = re.search(r"startup seconds (\{.*\})", p.stdout)
            res[tag] = {"images_per_s_file_to_file": float(m.group(3)) if m else None, "loop_seconds": float(m.group(2)) if m else None,
                        "main_thread_seconds": json.loads(mt.group(1).replace("'", '"')) if mt else None,
                        "startup_seconds": json.loads(su.group(1).replace("'", '"')) if su else None,
                        "process_wall_seconds_incl_checkpoint_load": round(wall, 2), "flags": " ".join(extra)}
        a = os.path.join(tmp, "res_sequential", name, "test_32", "images")
        lvl1 = os.path.join(tmp, "res_pipelined_procs_png_level_1", name, "test_32", "images")
        if os.path.isdir(a) and os.path.isdir(lvl1):      # other bytes, the same pixels
            fa = sorted(os.listdir(a))
            res["png_level_1_same_pixels"] = sorted(os.listdir(lvl1)) == fa and all(
                np.array_equal(np.asarray(Image.open(os.path.join(a, f))), np.asarray(Image.open(os.path.join(lvl1, f)))) for f in fa[::16])
            res["png_level_1_bytes_vs_default"] = round(sum(os.path.getsize(os.path.join(lvl1, f)) for f in fa) / max(1, sum(os.path.getsize(os.path.join(a, f)) for f in fa)), 3)
        for t in ("pipelined", "pipelined_procs", "sequential_decode_in_workers"):
            b = os.path.join(tmp, "res_" + t, name, "test_32", "images")
            if os.path.isdir(a) and os.path.isdir(b):
                fa, fb = sorted(os.listdir(a)), sorted(os.listdir(b))
                res["files_written"] = len(fb)
                res["byte_identical_" + t] = fa == fb and all(open(os.path.join(a, f), "rb").read() == open(os.path.join(b, f), "rb").read() for f in fa)
        res["byte_identical"] = all(v for k, v in res.items() if k.startswith("byte_identical_"))
        try:
            res["cpu_quota"] = open("/sys/fs/cgroup/cpu.max").read().strip()
        except OSError:
            res["cpu_quota"] = None
        res["usable_cpus_sched_getaffinity"] = len(os.sched_getaffinity(0))
        pipe = res.get("pipelined", {}).get("images_per_s_file_to_file")
        if pipe:
            res["bound"] = ("I/O bound, on the host: PNG encode (%.0f images/s on %d writer threads) and decode (%.0f images/s on %d workers) bound the file-to-file rate "
                            "(the pipelined loop's clock starts at its first batch: the workers, forked before the model is built, have decoded ahead while the checkpoint "
                        "loaded); the GPU alone replays these batches at bench.py's headline rate" % (res["encode_only_images_per_s"], writers, res["decode_only_images_per_s"], threads))
        print(json.dumps(res))
        if out_path:
            with open(out_path, "w") as f:
                json.dump(res, f, indent=1)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
