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

# secp256k1 field / group operations: SIMD time per wavefront-operation at 1..5 resident waves
# per SIMD (1024 one-wave blocks = one wave per SIMD), in units of the v_mad_u64_u32 issue time
mad_rate = out["v_mad_u64_u32"] * 1e12 / (1024 * 64)       # wave-mads per second per SIMD
fnames = {10: "field mul", 11: "field sqr", 12: "2 interleaved muls", 13: "add+sub", 14: "jacobian dbl",
          15: "mixed add", 16: "wide product only", 17: "wide square only", 18: "reduction only"}
table = {}
for kind, name in fnames.items():
    row = []
    for w in (1, 2, 3, 4, 5, 6, 8):
        ms, ops = ctx.probe_valu(kind, 1024 * w, 2000)
        per_simd = ops / 64 / 1024                            # wave-operations each SIMD executed
        t = ms * 1e-3 / per_simd                              # SIMD seconds per wave-operation
        row.append(round(t * mad_rate, 1))
    table[name] = row
    print("%-20s SIMD time per op, in mad-issue units, at 1,2,3,4,5,6,8 waves/SIMD: %s" % (name, row))
print(json.dumps(table))
