#!/usr/bin/env python3
"""A/B timing of several builds of the library on ONE GPU box (boxes of the pool differ by ~5 %,
so variants are only comparable inside one call).  Developer tool.

    python tools/ab_variants.py lib_a.so lib_b.so ...        # on the GPU box

Every variant runs in its own process (ELLGPU_LIB) on the same 2^20 secp256k1 verify tuples
(bench.py's generator, cached in /tmp), is checked against the expected mask and against the
first variant's P*k / G*k bytes, and reports the HIP-event kernel times."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = "/tmp/ab_inputs.npz"


def child(lib, n, reps):
    import numpy as np
    import torch
    import elliptic_amd
    import bench
    ctx = elliptic_amd.Context(0)
    if not os.path.exists(CACHE):
        h, r, s, pub, ok = bench.make_signatures(ctx, n, "ellgpu-ab")
        ks = bench.xof("ab:k", n * 32).reshape(n, 32).copy()
        np.savez(CACHE, h=h, r=r, s=s, pub=pub, ok=ok, ks=ks)
    d = np.load(CACHE)
    dev = torch.device("cuda", 0)
    dh, dr, ds_, dq, dk = (torch.from_numpy(d[k]).to(dev) for k in ("h", "r", "s", "pub", "ks"))
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = {"lib": os.path.basename(lib)}
    ctx.reserve("secp256k1", n)
    for _ in range(3):
        ctx.ecdsa_verify_dev("secp256k1", dh, dr, ds_, dq, dok)
    torch.cuda.synchronize()
    out["mask_ok"] = bool(np.array_equal(dok.cpu().numpy(), d["ok"]))
    ctx.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.ecdsa_verify_dev("secp256k1", dh, dr, ds_, dq, dok)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = ctx.get_timing()
    ctx.set_timing(False)
    out["verify_Mps"] = n * reps / dt / 1e6
    for k, (c, ms) in tm.items():
        out[k + "_ms"] = round(ms / max(c, 1), 4)
    # P*k and G*k -> affine bytes, compared across variants
    oxy = torch.zeros(n, 64, dtype=torch.uint8, device=dev)
    oinf = torch.zeros(n, dtype=torch.uint8, device=dev)
    for name, fn in (("mul_var", lambda: ctx.mul_var_dev("secp256k1", dk, dq, oxy, oinf)),
                     ("mul_fixed", lambda: ctx.mul_fixed_dev("secp256k1", dk, oxy, oinf))):
        fn()
        torch.cuda.synchronize()
        ref = "/tmp/ab_ref_%s.npy" % name
        got = oxy.cpu().numpy()
        if os.path.exists(ref):
            out[name + "_same"] = bool(np.array_equal(np.load(ref), got))
        else:
            np.save(ref, got)
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(max(reps // 3, 3)):
            fn()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = ctx.get_timing()
        ctx.set_timing(False)
        out[name + "_Mps"] = n * max(reps // 3, 3) / dt / 1e6
        c, ms = tm.get(name, (0, 0.0))
        out[name + "_kernel_ms"] = round(ms / max(c, 1), 4)
        for k2, (c2, ms2) in tm.items():
            if k2 != name:
                out["%s/%s_ms" % (name, k2)] = round(ms2 / max(c2, 1), 4)
    # field-layer probes at 3 and 4 waves / SIMD (units of one v_mad_u64_u32 issue)
    try:
        best = 0
        for _ in range(3):
            ms, ops = ctx.probe_valu(0, 256 * 8 * 4, 4096)
            best = max(best, ops / (ms * 1e-3))
        rate = best / (1024 * 64)
        out["mad_T_per_s"] = round(best / 1e12, 2)
        for kind, nm in ((10, "mul"), (11, "sqr"), (13, "addsub"), (14, "dbl"), (15, "madd")):
            row = []
            for w in (3, 4):
                ms, ops = ctx.probe_valu(kind, 1024 * w, 2000)
                row.append(round(ms * 1e-3 / (ops / 64 / 1024) * rate, 1))
            out["u_" + nm] = row
    except Exception as e:      # older builds
        out["probe_error"] = str(e)
    print(json.dumps(out), flush=True)
    ctx.close()


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
        return
    libs = [a for a in sys.argv[1:] if not a.startswith("--")]
    n = 1 << 20
    reps = 30
    for a in sys.argv[1:]:
        if a.startswith("--n="):
            n = int(a[4:])
        if a.startswith("--reps="):
            reps = int(a[7:])
    for f in (CACHE, "/tmp/ab_ref_mul_var.npy", "/tmp/ab_ref_mul_fixed.npy"):
        if os.path.exists(f):
            os.remove(f)
    passes = 1
    for a in sys.argv[1:]:
        if a.startswith("--passes="):
            passes = int(a[9:])
    order = libs * passes + libs[:1]       # the first variant once more at the end: drift / noise
    if True:
        for lib in order:
            env = dict(os.environ, ELLGPU_LIB=os.path.abspath(lib))
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, str(n), str(reps)],
                               env=env, capture_output=True, text=True, timeout=600)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            print(line[-1] if line else json.dumps({"lib": lib, "error": (p.stderr or p.stdout)[-600:]}), flush=True)


if __name__ == "__main__":
    main()
