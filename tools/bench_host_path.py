#!/usr/bin/env python3
"""PCIe-inclusive throughput of the host-buffer entry points (what the N-API addon and any
other FFI caller sees): ellgpu_ecdsa_verify / ellgpu_mul_var on pageable numpy buffers,
H2D + kernels + D2H inside the timed region.  GPU box only; never bench.py's `value`.

    python tools/bench_host_path.py [--n 1048576] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import elliptic_amd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    ctx = elliptic_amd.Context(0)
    hz, hr, hs, hq, want = bench.make_signatures(ctx, a.n, "host-path")

    def timed(name, fn, check):
        out = fn()
        assert check(out), name + ": parity"
        t = []
        for _ in range(a.reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        best, mean = min(t), sum(t) / len(t)
        print(json.dumps({"config": name, "n": a.n, "items_per_s_best": a.n / best,
                          "items_per_s_mean": a.n / mean, "ms_best": best * 1e3}), flush=True)

    timed("secp256k1 ecdsa_verify, host buffers (H2D 160 B + D2H 1 B per item)",
          lambda: ctx.ecdsa_verify("secp256k1", hz, hr, hs, hq),
          lambda ok: np.array_equal(np.asarray(ok).astype(np.uint8), want.astype(np.uint8)))
    # the same batch in wire formats: DER signatures + compressed keys (H2D 32 + 72 + 4 + 33 B)
    packed = ctx._pack_records(ctx.sig_to_der("secp256k1", hr, hs))
    keys = ctx.encode_points("secp256k1", hq, compact=True)
    timed("secp256k1 ecdsa_verify_wire, host buffers (DER + compressed keys, H2D %d B + D2H 2 B per item)"
          % (32 + packed[0].shape[1] + 4 + 33),
          lambda: ctx.ecdsa_verify_wire("secp256k1", hz, packed, keys),
          # (a corrupted key's x may have no y at all: err = 'invalid point', still not ok)
          lambda o: np.array_equal(o[0], want.astype(np.uint8)) and not o[1][want.astype(bool)].any())
    xy0, inf0 = ctx.mul_var("secp256k1", hr, hq)
    timed("secp256k1 mul_var, host buffers (H2D 96 B + D2H 65 B per item)",
          lambda: ctx.mul_var("secp256k1", hr, hq),
          lambda o: np.array_equal(o[0], xy0) and np.array_equal(o[1], inf0))
    fx0, fi0 = ctx.mul_fixed("secp256k1", hr)
    fb = (np.zeros_like(fx0), np.zeros_like(fi0))
    timed("secp256k1 mul_fixed, host buffers (H2D 32 B + D2H 65 B per item), result buffers reused",
          lambda: ctx.mul_fixed("secp256k1", hr, out=fb),
          lambda o: np.array_equal(o[0], fx0) and np.array_equal(o[1], fi0))
    bufs = (np.zeros_like(xy0), np.zeros_like(inf0))
    timed("secp256k1 mul_var, host buffers, result buffers reused",
          lambda: ctx.mul_var("secp256k1", hr, hq, out=bufs),
          lambda o: np.array_equal(o[0], xy0) and np.array_equal(o[1], inf0))


if __name__ == "__main__":
    main()
