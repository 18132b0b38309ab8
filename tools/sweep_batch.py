#!/usr/bin/env python3
"""Variable-base P*k throughput against batch size (developer tool, GPU box): a kernel whose
registers admit w waves per SIMD runs a batch of n items in rounds of w x 65 536 lanes; a batch
that leaves a partial last round finishes with under-occupied SIMDs (a lone wave issues at half
rate).  Shows where BASELINE's batch sizes sit on that staircase.

    python tools/sweep_batch.py p384 196608,262144,393216,524288 [--reps 6]
"""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import elliptic_amd


def rnd(seed, n, w):
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def main():
    curve = sys.argv[1]
    sizes = [int(x) for x in sys.argv[2].split(",")]
    reps = int(sys.argv[sys.argv.index("--reps") + 1]) if "--reps" in sys.argv else 6
    ctx = elliptic_amd.Context(0)
    dev = torch.device("cuda", 0)
    B = elliptic_amd.FIELD_BYTES[curve]
    nmax = max(sizes)
    k = rnd("sweep:k:" + curve, nmax, B)
    d = rnd("sweep:d:" + curve, nmax, B)
    if curve == "p521":
        k[:, 0] &= 1
        d[:, 0] &= 1
    if curve == "ed25519":
        k[:, 0] &= 0x0F
        d[:, 0] &= 0x0F
    dk, dd = torch.from_numpy(k).to(dev), torch.from_numpy(d).to(dev)
    pts = torch.zeros((nmax, 2 * B), dtype=torch.uint8, device=dev)
    out = torch.zeros((nmax, 2 * B), dtype=torch.uint8, device=dev)
    inf = torch.zeros(nmax, dtype=torch.uint8, device=dev)
    ctx.mul_fixed_dev(curve, dd, pts, inf)
    torch.cuda.synchronize()
    for n in sizes:
        fn = lambda: ctx.mul_var_dev(curve, dk[:n], pts[:n], out[:n], inf[:n])      # noqa: E731
        fn()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tm = ctx.get_timing()
        ctx.set_timing(False)
        print(json.dumps({"curve": curve, "n": n, "waves_per_simd": n / 65536.0, "M_items_per_s": n / dt / 1e6,
                          "ms": dt * 1e3, "kernels_ms": {kk: round(v[1] / v[0], 4) for kk, v in tm.items()},
                          "lib": os.path.basename(os.environ.get("ELLGPU_LIB", "default"))}), flush=True)


if __name__ == "__main__":
    main()
