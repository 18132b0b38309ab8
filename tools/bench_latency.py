#!/usr/bin/env python3
"""Latency of small batches through the host-buffer entry points (what one patched
`ec.verify` / `point.mul` call or a small batch costs).  GPU box only.

    python tools/bench_latency.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import elliptic_amd


def main():
    ctx = elliptic_amd.Context(0)
    n_max = 1 << 16
    hz, hr, hs, hq, want = bench.make_signatures(ctx, n_max, "latency")
    # ELLGPU_LATENCY_SIZES="512,2048,...": other batch sizes (the A/B of the one-item-per-row layer sets
    # ELLGPU_ROW_GRID=0 beside it: that form off)
    sizes = [int(x) for x in os.environ.get("ELLGPU_LATENCY_SIZES", "1,16,64,256,1024,2048,4096,8192,16384,65536").split(",")]
    for n in sizes:
        z, r, s, q = hz[:n].copy(), hr[:n].copy(), hs[:n].copy(), hq[:n].copy()
        for name, fn in (("ecdsa_verify", lambda: ctx.ecdsa_verify("secp256k1", z, r, s, q)),
                         ("mul_var", lambda: ctx.mul_var("secp256k1", r, q)),
                         ("mul_fixed", lambda: ctx.mul_fixed("secp256k1", r))):
            fn()
            ts = []
            for _ in range(30):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            print(json.dumps({"op": name, "n": n, "median_us": ts[len(ts) // 2] * 1e6, "min_us": ts[0] * 1e6,
                              "items_per_s": n / ts[len(ts) // 2]}), flush=True)


if __name__ == "__main__":
    main()
