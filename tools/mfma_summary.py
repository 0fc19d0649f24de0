#!/usr/bin/env python3
"""Per-kernel MFMA utilisation and HBM rate from rocprofv3 PMC passes of bench.py (kernel-trace + pmc only, separate runs):

  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_mfma \\
            -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --min-seconds 0
  rocprofv3 --kernel-trace --pmc FETCH_SIZE  ... -d gpurun_out/pmc_fetch -- (same command)
  rocprofv3 --kernel-trace --pmc WRITE_SIZE  ... -d gpurun_out/pmc_write -- (same command)
  python tools/mfma_summary.py <mfma counter_collection.csv> [<fetch csv> <write csv>] profiles/rNN

Formulas (MI355X_MICROARCH.md: rocprofv3 PMC slots, cycle constants, HBM section):
  kernel cycles      = GRBM_GUI_ACTIVE / 8                  (rocprofv3 reports the sum over the 8 XCDs)
  MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles * 256 CUs * 4 SIMDs)      (the counter sums busy cycles of every SIMD)
  executed MFMA rate = busy cycles * 1024 flop / duration   (v_mfma_f32_16x16x32_f16: 16384 flop per 16 busy cycles; includes padded lanes /
                       head_dim 24 -> 32 padding, so it is >= the algorithmic rate bench.py reports)
  HBM bytes          = 2 * FETCH_SIZE KiB + WRITE_SIZE KiB  (gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads)
Output: <out>_mfma_util.csv (one row per kernel, per forward of batch 8) and, when the fetch / write passes are given,
<out>_pmc_hbm.csv + <out>_pmc_traffic.json (the latter feeds bench.py's roofline.traffic cross-reference)."""
import collections, csv, json, sys

HBM_PEAK_GBS = 8000.0
MFMA_PEAK_TF = 2500.0


def short(name):
    """"_ZN12_GLOBAL__N_15k_mlpIDF16_Li12ELi2E...E" / "(anonymous namespace)::k_conv7_tz(...)" -> "k_mlp<f16,12,2,...>": kernel + template arguments"""
    import re
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", name)
    if not m:      # already demangled by the profiler: "void (anonymous namespace)::k_mlp2<12, 2, 4, 32>(Grouped<MlpArgs>)"
        d = name.replace("(anonymous namespace)::", "").replace("void ", "")
        depth, cut = 0, len(d)
        for i, ch in enumerate(d):
            depth += ch == "<"
            depth -= ch == ">"
            if ch == "(" and depth == 0:
                cut = i
                break
        return d[:cut].replace(" ", "")[:72]
    n = int(m.group(1))
    base, rest = name[m.end():m.end() + n], name[m.end() + n:]
    args = []
    if rest.startswith("I"):
        i = 1
        while i < len(rest) and rest[i] != "E":
            if rest.startswith("DF16_", i):
                args.append("f16"); i += 5
            elif rest[i] == "f":
                args.append("f32"); i += 1
            elif rest[i] == "L":
                j = rest.index("E", i)
                args.append(rest[i + 2:j].replace("n", "-")); i = j + 1
            else:
                break
    return base + ("<" + ",".join(args) + ">" if args else "")


# bench.py kernel key (block kind + level : step) -> the kernel instantiation that runs it, where that is one-to-one
BENCH_KEY = (("lvit3:proj_mlp_stream", "k_mlp3<24,"), ("lvit3:front_stream", "k_front3<"), ("lvit1:window_block_fused", "k_lvit_window<"), ("lvit1:proj_mlp_fused", "k_mlp2<6,"), ("lvit2:proj_mlp_fused", "k_mlp2<12,"),
             ("lvit1:embed_ln_qkv", "k_embed_qkv2<6,"), ("lvit2:embed_ln_qkv", "k_embed_qkv2<12,"))


def bench_kernels(rows):
    out = {}
    for key, frag in BENCH_KEY:
        for r in rows:
            if r["kernel"].startswith(frag) and r["hbm_bytes_per_launch"] is not None:
                out[key] = {"kernel": r["kernel"], "hbm_bytes_per_launch": round(r["hbm_bytes_per_launch"])}
    return out


def load(path):
    """-> {kernel: {"n": dispatches, "ns": total duration, counter: total}}; every dispatch appears once per counter"""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        key = (r["Dispatch_Id"], k)
        if key not in seen:
            seen.add(key)
            per[k]["n"] += 1
            per[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return per


def main():
    args = sys.argv[1:]
    out = args[-1]
    mf = load(args[0])
    fetch = load(args[1]) if len(args) >= 4 else {}
    write = load(args[2]) if len(args) >= 4 else {}
    ours = [k for k in mf if "k_" in k and ("_GLOBAL__N_" in k or "anonymous" in k)]
    # forwards in the trace = dispatches of a kernel that runs exactly once per forward
    once = [k for k in ours if "nchw_to_nhwc" in k or "u8hwc_to_nhwc" in k or "k_head5" in k]   # (round 4: k_head5 reads the network input itself)
    nfwd = int(sum(mf[k]["n"] for k in once)) or 1
    rows = []
    for k in ours:
        m = mf[k]
        n, us = m["n"], m["ns"] / 1e3
        cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        busy = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        util = busy / (cyc * 1024.0) if cyc else 0.0
        exe_tf = busy * 1024.0 / (us * 1e6) if us else 0.0          # flop / us / 1e6 = TFLOP/s
        fb = 2.0 * 1024.0 * fetch[k]["FETCH_SIZE"] / fetch[k]["n"] if k in fetch and fetch[k]["n"] else None
        wb = 1024.0 * write[k]["WRITE_SIZE"] / write[k]["n"] if k in write and write[k]["n"] else None
        hbm = (fb or 0.0) + (wb or 0.0) if fb is not None else None
        gbs = hbm / (us / n * 1e3) if hbm is not None and us else None      # bytes / ns = GB/s
        f_m, f_h = exe_tf / MFMA_PEAK_TF, (gbs / HBM_PEAK_GBS if gbs is not None else 0.0)
        rows.append(dict(kernel=short(k), launches_per_forward=n / nfwd, avg_us=us / n, us_per_forward=us / nfwd, mfma_busy_pct=100 * util,
                         executed_mfma_tflops=exe_tf, clock_ghz=(cyc / (us * 1e3) if us else 0.0),
                         hbm_bytes_per_launch=hbm, hbm_gbs=gbs, bound="mfma" if f_m >= f_h else "hbm", roof_frac=max(f_m, f_h), raw=k))
    rows.sort(key=lambda r: -r["us_per_forward"])
    tot = sum(r["us_per_forward"] for r in rows)
    with open(out + "_mfma_util.csv", "w") as f:
        f.write("kernel,launches_per_forward,avg_us,us_per_forward,share_pct,mfma_busy_pct,executed_mfma_tflops,clock_ghz,hbm_bytes_per_launch,hbm_gbs,"
                "nearest_roof,frac_of_that_roof\n")
        for r in rows:
            f.write('"%s",%.1f,%.2f,%.1f,%.1f,%.2f,%.1f,%.2f,%s,%s,%s,%.3f\n' % (
                r["kernel"], r["launches_per_forward"], r["avg_us"], r["us_per_forward"], 100 * r["us_per_forward"] / tot, r["mfma_busy_pct"],
                r["executed_mfma_tflops"], r["clock_ghz"], "%.0f" % r["hbm_bytes_per_launch"] if r["hbm_bytes_per_launch"] is not None else "",
                "%.0f" % r["hbm_gbs"] if r["hbm_gbs"] is not None else "", r["bound"], r["roof_frac"]))
        wsum = sum(r["mfma_busy_pct"] * r["us_per_forward"] for r in rows) / tot
        f.write('"ALL KERNELS (time-weighted)",,,%.1f,100.0,%.2f,,,,,,\n' % (tot, wsum))
    print("%d forwards in the trace, %.1f us of kernel time per forward, time-weighted MFMA busy %.2f %%" % (nfwd, tot, wsum))
    for r in rows[:14]:
        print("%-58s x%-5.1f %8.1f us  mfma busy %5.1f %%  %7.0f TF  %s" % (r["kernel"], r["launches_per_forward"], r["us_per_forward"], r["mfma_busy_pct"],
                                                                        r["executed_mfma_tflops"], ("%.0f GB/s" % r["hbm_gbs"]) if r["hbm_gbs"] else ""))
    if fetch:
        with open(out + "_pmc_hbm.csv", "w") as f:
            f.write("kernel,launches,fetch_bytes_per_launch_corrected_x2,write_bytes_per_launch,hbm_bytes_per_launch\n")
            for r in rows:
                if r["hbm_bytes_per_launch"] is not None:
                    k = r["raw"]
                    f.write('"%s",%d,%.0f,%.0f,%.0f\n' % (r["kernel"], fetch[k]["n"], 2048.0 * fetch[k]["FETCH_SIZE"] / fetch[k]["n"],
                                                        1024.0 * write[k]["WRITE_SIZE"] / max(1.0, write[k]["n"]) if k in write else 0.0, r["hbm_bytes_per_launch"]))
        json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), fetch doubled per the gfx950 guide; keyed by kernel symbol",
                   "kernels": {r["kernel"]: {"hbm_bytes_per_launch": round(r["hbm_bytes_per_launch"])} for r in rows if r["hbm_bytes_per_launch"] is not None},
                   "bench_kernels": bench_kernels(rows)}, open(out + "_pmc_traffic.json", "w"), indent=1)


if __name__ == "__main__":
    main()
