#!/usr/bin/env python3
"""For one graph replay in a rocprofv3 kernel_trace.csv: time spent with 0/1/2/3+ kernels resident, and per kernel
name the time it ran ALONE (critical-path candidates) vs overlapped."""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:40]) for r in rows]
starts = [i for i, k in enumerate(ks) if "k_nchw_to_nhwc" in k[2]]
idx = int(sys.argv[2]) if len(sys.argv) > 2 else -3
a, b = starts[idx], starts[idx + 1]
seg = ks[a:b]
ev = []
for i, (s, e, n) in enumerate(seg):
    ev.append((s, 1, i)); ev.append((e, -1, i))
ev.sort()
active = set(); last = ev[0][0]
depth_time = collections.Counter(); alone = collections.Counter(); shared = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        depth_time[min(len(active), 4)] += dt
        for j in active:
            (alone if len(active) == 1 else shared)[seg[j][2]] += dt
    last = t
    if d > 0: active.add(i)
    else: active.discard(i)
tot = sum(depth_time.values())
print("wall %.0f us; kernels resident: " % (tot / 1e3) + ", ".join("%s: %.0f us (%.0f%%)" % (("4+" if k == 4 else k), v / 1e3, 100.0 * v / tot) for k, v in sorted(depth_time.items())))
print("%-42s %10s %10s" % ("kernel", "alone us", "shared us"))
for n in sorted(set(alone) | set(shared), key=lambda n: -(alone[n])):
    print("%-42s %10.0f %10.0f" % (n, alone[n] / 1e3, shared[n] / 1e3))
