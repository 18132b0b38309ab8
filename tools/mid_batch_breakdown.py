#!/usr/bin/env python3
"""Kernel times of ONE host-buffer call of a few hundred to a few thousand items (developer tool, GPU box):
where a mid-size batch -- the wave-per-item layer up to 640 items, one item per row up to 4 608, the
one-lane kernels above -- spends its time, HIP-event time per kernel beside the call's wall time.

    python tools/mid_batch_breakdown.py [n ...]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import elliptic_amd


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [64, 256, 640, 1024, 1366, 2048, 3072, 4096, 4608, 8192, 16384]
    ctx = elliptic_amd.Context(0)
    hz, hr, hs, hq, want = bench.make_signatures(ctx, max(sizes), "latency")
    for op in ("ecdsa_verify", "mul_var"):
        for n in sizes:
            z, r, s, q = (a[:n].copy() for a in (hz, hr, hs, hq))
            fn = (lambda: ctx.ecdsa_verify("secp256k1", z, r, s, q)) if op == "ecdsa_verify" else (lambda: ctx.mul_var("secp256k1", r, q))
            for _ in range(3):
                fn()
            ctx.set_timing(True)
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            tm = ctx.get_timing()
            ctx.set_timing(False)
            ts.sort()
            print(json.dumps({"op": op, "n": n, "call_median_us": round(ts[len(ts) // 2] * 1e6, 1),
                              "kernels_us": {k: round(v[1] / max(v[0], 1) * 1e3, 1) for k, v in tm.items()},
                              "launches_per_call": round(sum(v[0] for v in tm.values()) / 30, 1)}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
