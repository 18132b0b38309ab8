#!/usr/bin/env python3
"""A/B of the parted verify (three lanes per item, engine.h FnEcdsaParts; three WAVES per item on
the row layer, FnEcdsaPartsC) against the one-lane ladder at small batch sizes, on ONE GPU box (developer tool; the override is read when a context
is created, so each leg is its own process).

    python tools/parted_ab.py [--sizes=1,64,1024,...] [--reps=200]

Per size: the device-resident pass (ecdsa_verify_dev, HIP-event kernel times + wall time per
pass) and the host-buffer call a patched EC#verify makes (median / best of 200 calls); the mask
is checked against the expected one on every leg."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SIZES = [1, 16, 64, 128, 256, 341, 512, 682, 1024, 1365, 2048, 4096, 16384, 32768, 65536]


def child(sizes, reps):
    import numpy as np
    import torch
    import bench
    import elliptic_amd
    ctx = elliptic_amd.Context(0)
    n0 = max(sizes)
    h, r, s, pub, expect = bench.cached_signatures(ctx, n0, "ellgpu-bench-v1:3:rank0")
    dev = torch.device("cuda", 0)
    dh, dr, dsg, dq = (torch.from_numpy(x).to(dev) for x in (h, r, s, pub))
    ctx.reserve("secp256k1", n0)
    grid = os.environ.get("ELLGPU_PARTED_GRID")
    coop = os.environ.get("ELLGPU_COOP_GRID")
    for n in sizes:
        dok = torch.zeros(n, dtype=torch.uint8, device=dev)
        args = (dh[:n], dr[:n], dsg[:n], dq[:n], dok)
        for _ in range(5):
            ctx.ecdsa_verify_dev("secp256k1", *args)
        torch.cuda.synchronize()
        ok = bool(np.array_equal(dok.cpu().numpy(), expect[:n]))
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            ctx.ecdsa_verify_dev("secp256k1", *args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tm = ctx.get_timing()
        ctx.set_timing(False)
        out = {"lib": os.path.basename(os.environ.get("ELLGPU_LIB", "default")), "parted_grid": grid, "coop_grid": coop, "n": n, "mask_ok": ok, "pass_ms": round(dt * 1e3, 4),
               "kernels_ms": {k: round(v[1] / max(v[0], 1), 4) for k, v in tm.items()}}
        if n <= 4096:
            z, rr, ss, q = h[:n].copy(), r[:n].copy(), s[:n].copy(), pub[:n].copy()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                got = ctx.ecdsa_verify("secp256k1", z, rr, ss, q)
                ts.append(time.perf_counter() - t0)
            ts.sort()
            out["host_call_ok"] = bool(np.array_equal(np.asarray(got), expect[:n]))
            out["host_call_median_us"] = round(ts[len(ts) // 2] * 1e6, 1)
            out["host_call_min_us"] = round(ts[0] * 1e6, 1)
        print(json.dumps(out), flush=True)


def main():
    sizes, reps = SIZES, 200
    for a in sys.argv[1:]:
        if a.startswith("--sizes="):
            sizes = [int(x) for x in a[8:].split(",")]
        elif a.startswith("--reps="):
            reps = int(a[7:])
    if "--child" in sys.argv:
        child(sizes, reps)
        return
    # three forms of a small batch: the whole ladder on one lane; the parts one item per lane; the
    # parts one item per WAVE (the row layer, csrc/coop.h) -- each leg its own process
    big = str(1 << 30)
    for grid, coop in (("0", "0"), (big, "0"), (big, big)) * (1 if "--once" in sys.argv else 2):
        env = dict(os.environ, ELLGPU_PARTED_GRID=grid, ELLGPU_COOP_GRID=coop)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--sizes=" + ",".join(map(str, sizes)),
                            "--reps=%d" % reps], env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print("\n".join(lines) if lines else json.dumps({"parted_grid": grid, "error": (p.stderr or p.stdout)[-800:]}), flush=True)


if __name__ == "__main__":
    main()
