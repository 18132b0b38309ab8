#!/usr/bin/env python3
"""GPU box: what ONE item's critical path pays per field / group operation -- the one-item-per-lane
secp256k1 field (fp.h; probe kinds 10 / 14 / 15 of ellgpu_probe_valu) against the lanes-per-item
layer (coop.h; kinds 20 / 24 / 25), on a lone wave (blocks = 1) and on 256 one-wave workgroups.

  python tools/microbench/row_probe.py > profiles/rNN_row_probe.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import elliptic_amd

ctx = elliptic_amd.Context(0)
out = {}
for blocks in (1, 256):
    row = {}
    for name, kind, iters in (("one_lane_mul", 10, 20000), ("one_lane_dbl", 14, 3000), ("one_lane_madd", 15, 3000),
                              ("row_mul", 20, 20000), ("row_dbl", 24, 3000), ("row_madd", 25, 3000)):
        best = min(ctx.probe_valu(kind, blocks, iters)[0] for _ in range(3))
        row[name] = round(best * 1e6 / iters, 1)
    out[str(blocks)] = row
print(json.dumps({"ns_per_op": out}))
