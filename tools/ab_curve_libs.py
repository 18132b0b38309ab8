#!/usr/bin/env python3
"""A/B of single-curve developer builds on ONE GPU box (boxes of the pool differ by ~5 %).

  build (container):   python tools/ab_curve_libs.py build CvP384 pair0=-DELL_SOLINAS_PAIR=0 pair1=-DELL_SOLINAS_PAIR=1
                       -> ab_libs/CvP384_pair0.so, ab_libs/CvP384_pair1.so   (git-ignored; they travel with gpurun)
  run (GPU box):       python tools/ab_curve_libs.py run p384 262144 ab_libs/CvP384_pair0.so ab_libs/CvP384_pair1.so

Every variant runs in its own process (ELLGPU_LIB): P*k on n seeded (scalar, point) pairs, one and
two passes in flight, HIP-event kernel times, bytes compared with the first variant's."""
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(curve, variants):
    from elliptic_amd import build as b
    os.makedirs(os.path.join(ROOT, "ab_libs"), exist_ok=True)
    base = list(b.FLAGS)
    for v in variants:
        name, flags = v.split("=", 1)
        b.FLAGS[:] = base + flags.split(",")
        out = b.build_dev_k256(curve=curve)
        dst = os.path.join(ROOT, "ab_libs", "%s_%s.so" % (curve, name))
        shutil.copy(out, dst)
        print("->", dst)
    b.FLAGS[:] = base


def child(curve, n, reps):
    import numpy as np
    import torch
    import elliptic_amd
    B = elliptic_amd.FIELD_BYTES[curve]
    ctx = elliptic_amd.Context(0)
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(5)
    k = rng.integers(0, 256, (n, B), dtype=np.uint8)
    d = rng.integers(0, 256, (n, B), dtype=np.uint8)
    if curve == "p521":
        k[:, 0] &= 1
        d[:, 0] &= 1
    dk, dd = torch.from_numpy(k).to(dev), torch.from_numpy(d).to(dev)
    pts = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
    inf = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.mul_fixed_dev(curve, dd, pts, inf)
    outs = [(torch.zeros_like(pts), torch.zeros_like(inf)) for _ in range(2)]
    lanes = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    res = {"lib": os.path.basename(os.environ.get("ELLGPU_LIB", "default")), "curve": curve, "n": n}
    for flight in (1, 2):
        def one(i):
            j = i % flight
            with torch.cuda.stream(lanes[j]):
                ctx.mul_var_dev(curve, dk, pts, outs[j][0], outs[j][1])
        for i in range(4):
            one(i)
        torch.cuda.synchronize()
        if flight == 1:
            ctx.set_timing(True)
        t0 = time.perf_counter()
        for i in range(reps):
            one(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        if flight == 1:
            tm = ctx.get_timing()
            ctx.set_timing(False)
            res["kernels_ms"] = {kk: round(v[1] / max(v[0], 1), 4) for kk, v in tm.items()}
        res["in_flight_%d" % flight] = {"ms_per_pass": round(dt * 1e3, 4), "M_items_per_s": round(n / dt / 1e6, 3)}
    got = outs[0][0].cpu().numpy()
    ref = "/tmp/ab_curve_ref_%s_%d.npy" % (curve, n)
    if os.path.exists(ref):
        res["same_bytes_as_first"] = bool(np.array_equal(np.load(ref), got))
    else:
        np.save(ref, got)
        res["same_bytes_as_first"] = True
    print(json.dumps(res), flush=True)


def main():
    if sys.argv[1] == "build":
        return build(sys.argv[2], sys.argv[3:])
    if sys.argv[1] == "child":
        return child(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]))
    curve, n, libs = sys.argv[2], int(sys.argv[3]), sys.argv[4:]
    for f in os.listdir("/tmp"):
        if f.startswith("ab_curve_ref_"):
            os.remove(os.path.join("/tmp", f))
    for rnd in range(2):                                   # twice, interleaved: clock drift shows as a difference between rounds
        for lib in libs:
            env = dict(os.environ, ELLGPU_LIB=os.path.abspath(lib))
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", curve, str(n), "12"], env=env, check=True)


if __name__ == "__main__":
    main()
