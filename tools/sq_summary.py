#!/usr/bin/env python3
"""Per-kernel SQ wave-state shares from one rocprofv3 --pmc pass (SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU):  python tools/sq_summary.py <counter_collection.csv>
WAIT_ANY (parked on s_waitcnt / barrier) + WAIT_INST_ANY (issue stall) + ACTIVE_INST_ANY ~= WAVE_CYCLES (MI355X_MICROARCH.md, PMC slots)."""
import collections, csv, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from mfma_summary import short
per = collections.defaultdict(lambda: collections.defaultdict(float))
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]
    if "k_" not in k:
        continue
    per[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if (r["Dispatch_Id"], k) not in seen:
        seen.add((r["Dispatch_Id"], k)); per[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); per[k]["n"] += 1
print("%-44s %6s %9s %7s %7s %7s %7s %8s" % ("kernel", "n", "us/launch", "wait%", "istall%", "active%", "ldsst%", "bankcf%"))
for k, m in sorted(per.items(), key=lambda kv: -kv[1]["ns"])[:24]:
    wc = m["SQ_WAVE_CYCLES"] or 1.0
    print("%-44s %6d %9.1f %7.1f %7.1f %7.1f %7.1f %8.2f" % (short(k)[:44], m["n"], m["ns"] / m["n"] / 1e3, 100 * m["SQ_WAIT_ANY"] / wc, 100 * m["SQ_WAIT_INST_ANY"] / wc,
                                                         100 * m["SQ_ACTIVE_INST_ANY"] / wc, 100 * m["SQ_WAIT_INST_LDS"] / wc, 100 * m["SQ_LDS_BANK_CONFLICT"] / max(1.0, m["SQ_BUSY_CYCLES"])))
