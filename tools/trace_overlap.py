#!/usr/bin/env python3
"""From a rocprofv3 kernel_trace.csv of bench.py: for the last graph replays, wall time per forward, union of kernel
intervals (GPU busy), sum of kernel durations (work), and the idle share."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# forwards are delimited by the input layout kernel
starts = [i for i, k in enumerate(ks) if "k_nchw_to_nhwc" in k[2]]
print("forwards found:", len(starts))
for a, b in list(zip(starts, starts[1:]))[-6:-1]:
    seg = [k for k in ks[a:b] if "GLOBAL__N_1" in k[2] or "k_" in k[2]]
    t0, t1 = seg[0][0], max(k[1] for k in seg)
    work = sum(k[1] - k[0] for k in seg)
    busy, cur_s, cur_e = 0, None, None
    for s, e, _ in sorted(seg):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    gaps = sorted(((s2 - e1) for (s1, e1, _), (s2, e2, _) in zip(sorted(seg), sorted(seg)[1:])), reverse=True)
    print("kernels %d  wall %.0f us  busy(union) %.0f us  idle %.0f us (%.0f%%)  work(sum) %.0f us  overlap factor %.2f" %
          (len(seg), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, 100.0 * (t1 - t0 - busy) / (t1 - t0), work / 1e3, work / busy))
