#!/usr/bin/env python3
"""One item per DPP ROW against one item per WAVE (round-5 review, "Next round" 2, step 1; GPU box).

The lanes-per-item layer gives a wave one item (coop.h FpK256C: the wave's four 16-lane rows hold
the same element, or four elements of that item inside a step of the group law).  FpK256R gives
each row an item of its own -- four items per wave, every product the row_newbcast instruction
stream, the group law's products one after the other.  This driver times dependent chains of
products, Jacobian doublings and mixed additions of both forms through ellgpu_probe_valu (kinds
20 / 24 / 25 and 30 / 34 / 35) at EQUAL GRIDS (waves per SIMD), checks that both forms compute the
same field values (ELLGPU_PROBE_DIGEST: a digest of the canonical result words of the first items),
and prints items per second and the ratio the review's gate is about (>= 2.5x).

  python tools/microbench/row_items.py [out.jsonl]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    import elliptic_amd
    ctx = elliptic_amd.Context(0)
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None
    rows = []
    # same values?  the first `b` items of a one-item-per-row launch of b / 4 waves are the items of b
    # one-item-per-wave waves
    os.environ["ELLGPU_PROBE_DIGEST"] = "1"
    for kc, kr, name in ((20, 30, "product"), (24, 34, "doubling"), (25, 35, "mixed addition")):
        for iters in (1, 2, 7):
            _, dc = ctx.probe_valu(kc, 64, iters)
            _, dr = ctx.probe_valu(kr, 16, iters)
            row = {"check": name, "iters": iters, "digest_item_per_wave": dc, "digest_item_per_row": dr, "equal": dc == dr}
            rows.append(row)
            print(json.dumps(row), flush=True)
            assert dc == dr, "the one-item-per-row field computes other values: %s" % name
    del os.environ["ELLGPU_PROBE_DIGEST"]
    iters = 2000
    for waves_per_simd in (0.25, 1, 2, 4):
        blocks = int(1024 * waves_per_simd)
        for kc, kr, name in ((20, 30, "product"), (24, 34, "doubling"), (25, 35, "mixed addition")):
            best = {}
            for kind in (kc, kr):
                ms = min(ctx.probe_valu(kind, blocks, iters)[0] for _ in range(3))
                best[kind] = ms
            per_wave_c = best[kc] * 1e6 / iters            # ns per operation of a wave's chain
            per_wave_r = best[kr] * 1e6 / iters
            row = {"op": name, "waves": blocks, "waves_per_simd": waves_per_simd,
                   "ns_per_op_item_per_wave": per_wave_c, "ns_per_op_item_per_row": per_wave_r,
                   "items_per_s_item_per_wave": blocks * iters / (best[kc] * 1e-3),
                   "items_per_s_item_per_row": 4 * blocks * iters / (best[kr] * 1e-3),
                   "ratio_items_per_s": 4 * best[kc] / best[kr]}
            rows.append(row)
            print(json.dumps(row), flush=True)
    if out:
        for r in rows:
            out.write(json.dumps(r) + "\n")
        out.close()
    ctx.close()


if __name__ == "__main__":
    main()
