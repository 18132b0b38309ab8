#!/usr/bin/env python3
"""A/B of library variants (workgroup size: -DELL_BLOCK=64 against 128) on the operations a
change of the launch geometry touches, at a full grid and at grids of one to four waves per CU
(developer tool; one process per variant).

    python tools/wg_ab.py lib_a.so lib_b.so ...
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SIZES = [1 << 20, 196608, 131072, 65536, 49152, 32768, 16384]


def child():
    import numpy as np
    import torch
    import bench
    import elliptic_amd
    ctx = elliptic_amd.Context(0)
    n0 = max(SIZES)
    h, r, s, pub, expect = bench.cached_signatures(ctx, n0, "ellgpu-bench-v1:3:rank0")
    dev = torch.device("cuda", 0)
    dh, dr, dsg, dq = (torch.from_numpy(x).to(dev) for x in (h, r, s, pub))
    ctx.reserve("secp256k1", n0)
    lib = os.path.basename(os.environ.get("ELLGPU_LIB", "default"))
    if os.environ.get("ELLGPU_ONE_WAVE_GROUPS") is not None:
        lib += " ELLGPU_ONE_WAVE_GROUPS=" + os.environ["ELLGPU_ONE_WAVE_GROUPS"]
    for n in SIZES:
        reps = 20 if n > 300000 else 100
        dok = torch.zeros(n, dtype=torch.uint8, device=dev)
        xy = torch.zeros((n, 64), dtype=torch.uint8, device=dev)
        inf = torch.zeros(n, dtype=torch.uint8, device=dev)
        ops = {"verify": lambda: ctx.ecdsa_verify_dev("secp256k1", dh[:n], dr[:n], dsg[:n], dq[:n], dok),
               "mul_var": lambda: ctx.mul_var_dev("secp256k1", dr[:n], dq[:n], xy, inf),
               "mul_fixed": lambda: ctx.mul_fixed_dev("secp256k1", dr[:n], xy, inf)}
        out = {"lib": lib, "n": n}
        for name, fn in ops.items():
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            if name == "verify":
                out["mask_ok"] = bool(np.array_equal(dok.cpu().numpy(), expect[:n]))
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            out[name + "_ms"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        print(json.dumps(out), flush=True)


def main():
    if "--child" in sys.argv:
        child()
        return
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    for lib in libs * 2:
        env = dict(os.environ)
        if lib != "default":
            env["ELLGPU_LIB"] = os.path.abspath(lib)
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True, timeout=900)
        lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
        print("\n".join(lines) if lines else json.dumps({"lib": lib, "error": (p.stderr or p.stdout)[-800:]}), flush=True)


if __name__ == "__main__":
    main()
