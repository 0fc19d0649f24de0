#!/usr/bin/env python3
"""Median kernel duration of each run of consecutive identical dispatches in a rocprofv3 kernel_trace.csv
(pairs with tools/bench_gemm.py: one run per (shape, kernel variant))."""
import csv, sys, statistics
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
key = lambda r: (r["Kernel_Name"], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size"), r.get("Grid_Size_Y"))
run, last = [], None
def flush():
    if last and len(run) >= 10 and "gemm" in last[0]:
        name = last[0].split("(")[0].replace("_ZN12_GLOBAL__N_1", "")[:30]
        print("%-30s grid %-8s %-6s n=%-3d median %.1f us  min %.1f" % (name, last[1], last[2], len(run), statistics.median(run), min(run)))
for r in rows:
    k = key(r)
    if k != last:
        flush(); run = []; last = k
    run.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
flush()
