#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of libcfen_hip.so (the code-object metadata llvm-objdump --offloading extracts):
what decides how many workgroups of which kernels fit on a CU together.  Usage: python tools/kernel_resources.py [filter]"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    tmp = tempfile.mkdtemp(prefix="cfen_res_")
    try:
        so = shutil.copy(os.path.join(ROOT, "cfen_vit_dehazing_amd", "libcfen_hip.so"), os.path.join(tmp, "lib.so"))
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            for blk in notes.split("  - .agpr_count:")[1:]:
                get = lambda k: int((re.search(r"\." + k + r":\s+(\d+)", blk) or [0, "0"])[1])
                name = re.search(r"\.name:\s+(\S+)", blk).group(1)
                rows.append((name, int(blk.split()[0]), get("vgpr_count"), get("sgpr_count"), get("group_segment_fixed_size"),
                             get("private_segment_fixed_size"), get("vgpr_spill_count"), get("max_flat_workgroup_size")))
        names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
        print("%-86s %5s %5s %5s %7s %7s %5s" % ("kernel", "agpr", "vgpr", "sgpr", "lds", "scratch", "spill"))
        for r, dn in sorted(zip(rows, names), key=lambda t: t[1]):
            dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
            dn = re.sub(r"\(.*", "", dn)
            if flt in dn:
                print("%-86s %5d %5d %5d %7d %7d %5d" % ((dn[:86],) + r[1:7]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
