#!/usr/bin/env python3
"""A/B of library variants at several batch sizes on ONE GPU box (developer tool).

    python tools/ab_small.py lib_a.so lib_b.so ... [--sizes=1048576,131072] [--reps=40] [--env=ELLGPU_PREP_K=1]

Each variant runs in its own process (ELLGPU_LIB) on the first n tuples of bench.py's batch;
the mask is checked against the expected one; HIP-event kernel times and wall time per pass."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


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
    out = {"lib": os.path.basename(os.environ.get("ELLGPU_LIB", "default")), "prep_k": os.environ.get("ELLGPU_PREP_K")}
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
        out["n%d" % n] = {"mask_ok": ok, "pass_ms": round(dt * 1e3, 4),
                          **{k: round(v[1] / max(v[0], 1), 4) for k, v in tm.items()}}
    print(json.dumps(out), flush=True)


def main():
    sizes = [1 << 20, 1 << 17]
    reps = 40
    envs = {}
    libs = []
    for a in sys.argv[1:]:
        if a.startswith("--sizes="):
            sizes = [int(x) for x in a[8:].split(",")]
        elif a.startswith("--reps="):
            reps = int(a[7:])
        elif a.startswith("--env="):
            k, v = a[6:].split("=", 1)
            envs[k] = v
        elif a != "--child":
            libs.append(a)
    if "--child" in sys.argv:
        child(sizes, reps)
        return
    for lib in libs + libs[:1]:
        env = dict(os.environ, **envs)
        if lib != "default":
            env["ELLGPU_LIB"] = os.path.abspath(lib)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--sizes=" + ",".join(map(str, sizes)),
                            "--reps=%d" % reps], env=env, capture_output=True, text=True, timeout=900)
        line = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print(line[-1] if line else json.dumps({"lib": lib, "error": (p.stderr or p.stdout)[-800:]}), flush=True)


if __name__ == "__main__":
    main()
