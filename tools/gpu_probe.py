#!/usr/bin/env python3
"""Integer-VALU micro-benchmark on the GPU box: dependency-free instruction
streams on every CU (see ellgpu_probe_valu in include/ellgpu.h)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elliptic_amd

ctx = elliptic_amd.Context(0)
names = {0: "v_mad_u64_u32", 1: "v_mul_lo_u32+v_mul_hi_u32", 2: "v_mad_u32_u24", 3: "v_add_co+v_addc"}
out = {}
for kind, name in names.items():
    best = 0
    for blocks in (256 * 4, 256 * 8, 256 * 16, 256 * 32):
        ms, ops = ctx.probe_valu(kind, blocks, 4096)
        best = max(best, ops / (ms * 1e-3))
    out[name] = best / 1e12
    print("%-28s %.3f T inner-ops/s" % (name, best / 1e12))
print(json.dumps(out))
