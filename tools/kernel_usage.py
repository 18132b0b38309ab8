#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stdin or a
log file) into one line per kernel."""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    dn = subprocess.run(["c++filt", name], capture_output=True,
                        text=True).stdout.strip().replace("ell::", "")
    dn = re.sub(r"^void k_run<(.*)>\(.*$", r"\1", dn)[:70]

    def g(k):
        m = re.search(k + r": (\S+)", b)
        return m.group(1) if m else "?"
    print("%-72s VGPR=%-4s AGPR=%-3s SGPR=%-4s scratch=%-6s occ=%-2s LDS=%s" % (
        dn, g("VGPRs"), g("AGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
        g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
