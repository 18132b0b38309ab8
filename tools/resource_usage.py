#!/usr/bin/env python3
"""Distil csrc/_obj/resource_usage.log (python -m elliptic_amd.build --remarks --force) into one
line per kernel: registers, scratch, occupancy, LDS.

    python -m elliptic_amd.build --remarks --force && python tools/resource_usage.py > profiles/rNN_kernel_resource_usage.txt
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOG = os.path.join(ROOT, "elliptic_amd", "csrc", "_obj", "resource_usage.log")


def main():
    txt = open(LOG).read()
    rows = {}
    cur = None
    for line in txt.splitlines():
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            rows.setdefault(cur, {})
            continue
        m = re.search(r"remark:\s+(VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and cur:
            rows[cur][m.group(1).split()[0]] = int(m.group(2))
    names = list(rows)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    out = []
    for mangled, name in zip(names, dem):
        r = rows[mangled]
        m = re.search(r"k_run<ell::(.*)>\(", name)
        short = (m.group(1) if m else name).replace("ell::", "")
        out.append("%-72s VGPR=%-4d AGPR=%-3d SGPR=%-4d scratch=%-6d occ=%d  LDS=%d" % (
            short, r.get("VGPRs", 0), r.get("AGPRs", 0), r.get("TotalSGPRs", 0), r.get("ScratchSize", 0),
            r.get("Occupancy", 0), r.get("LDS", 0)))
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
