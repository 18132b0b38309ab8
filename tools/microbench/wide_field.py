#!/usr/bin/env python3
"""ns per operation of a dependent chain on the WIDE lanes-per-item layer (csrc/coop_wide.h: p384 /
p521, an element over the lanes of a wave) -- ellgpu_probe_valu kinds 40 / 44 / 45 (p384 product /
doubling / mixed addition) and 50 / 54 / 55 (p521), on a lone wave and at 1 and 2 waves per SIMD --
beside the one-item-per-lane chains they replace for a lone call.  GPU box.

  python tools/microbench/wide_field.py [out.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import elliptic_amd
    ctx = elliptic_amd.Context(0)
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    iters = 1000
    for blocks in (1, 1024, 2048):
        for kind, name in ((40, "p384 product"), (44, "p384 doubling (a = -3)"), (45, "p384 mixed addition"),
                           (50, "p521 product"), (54, "p521 doubling (a = -3)"), (55, "p521 mixed addition")):
            ms = min(ctx.probe_valu(kind, blocks, iters)[0] for _ in range(3))
            row = {"op": name, "waves": blocks, "ns_per_op": ms * 1e6 / iters}
            print(json.dumps(row), flush=True)
            if out:
                out.write(json.dumps(row) + "\n")
    ctx.close()


if __name__ == "__main__":
    main()
