#!/usr/bin/env python3
"""Per (kernel, grid) duration statistics from a rocprofv3 kernel_trace.csv:  kernel_trace_fold.py <csv> [--match substr]"""
import csv, sys, collections
f = sys.argv[1]
match = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--match" else None
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    k = row["Kernel_Name"]
    if match and match not in k:
        continue
    acc[(k[:80], row.get("Grid_Size_X", row.get("Grid_Size", "?")), row.get("Grid_Size_Y", ""), row.get("Grid_Size_Z", ""))].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
for (k, gx, gy, gz), d in sorted(acc.items()):
    d.sort()
    print("%-80s grid %s,%s,%s  x%-3d median %8.1f us  min %8.1f  max %8.1f" % (k, gx, gy, gz, len(d), d[len(d) // 2] / 1e3, d[0] / 1e3, d[-1] / 1e3))
