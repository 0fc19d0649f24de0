#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, kernel-trace only) into per-kernel HBM bytes.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph
  python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> profiles/rNN_x

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of
the bytes of wide coalesced reads, so fetch is doubled; WRITE_SIZE is exact.  Writes <out>_pmc_hbm.csv (per kernel) and
<out>_pmc_traffic.json (per bench.py kernel class: bytes per launch), which bench.py reports as roofline.traffic.
"""
import collections, csv, json, sys

CLASS_OF = (("k_gemm_nt", "gemm"), ("k_gemm_skinny", "gemm"), ("k_gemm", "gemm"), ("k_attention", "attention"), ("k_layernorm", "layernorm"),
            ("k_patchify", "tokens"), ("k_unpatchify", "tokens"), ("k_upsample4", "tokens"), ("k_nchw_to_nhwc", "tokens"),
            ("k_conv", "conv"), ("k_convT", "conv"), ("k_conv7", "conv"), ("k_chan_stats", "norm"), ("k_instnorm", "norm"), ("k_cfsm", "norm"), ("k_mlp", "mlp"), ("k_embed_qkv", "gemm"))


def fold(path, counter):
    tot = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            t = tot[r["Kernel_Name"]]
            t[0] += 1
            t[1] += float(r["Counter_Value"]) * 1024.0
    return tot


def main():
    fetch, write, out = fold(sys.argv[1], "FETCH_SIZE"), fold(sys.argv[2], "WRITE_SIZE"), sys.argv[3]
    classes = collections.defaultdict(lambda: [0, 0.0])
    with open(out + "_pmc_hbm.csv", "w") as f:
        f.write("kernel,launches,fetch_bytes_per_launch_corrected_x2,write_bytes_per_launch,hbm_bytes_per_launch\n")
        for k in sorted(fetch, key=lambda k: -fetch[k][1]):
            if "_GLOBAL__N_" not in k and "k_" not in k:
                continue
            n = fetch[k][0]
            fb = 2.0 * fetch[k][1] / n
            wb = write[k][1] / max(1, write[k][0]) if k in write else 0.0
            f.write('"%s",%d,%.0f,%.0f,%.0f\n' % (k, n, fb, wb, fb + wb))
            for frag, cls in CLASS_OF:
                if frag in k:
                    classes[cls][0] += n
                    classes[cls][1] += (fb + wb) * n
                    break
    js = {c: {"launches": v[0], "hbm_bytes_per_launch": round(v[1] / v[0])} for c, v in classes.items()}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), fetch doubled per gfx950 guide", "classes": js},
              open(out + "_pmc_traffic.json", "w"), indent=1)
    print(json.dumps(js, indent=1))


if __name__ == "__main__":
    main()
