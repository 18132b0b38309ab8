#!/usr/bin/env python3
"""P*k throughput on a user-defined short curve (brainpoolP256r1 through ellgpu_curve_define_short),
device-resident buffers, HIP-event timing.  Developer tool (GPU box).

    [ELLGPU_LIB=variant.so] python tools/bench_custom.py [log2 n]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import numpy as np
    import torch
    import elliptic_amd
    import bench
    import parity_checks as PC
    from golden_util import I
    n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 18)
    sp = [s for s in PC.custom_curves() if s["name"] == "brainpoolP256r1"][0]
    ctx = elliptic_amd.Context(0)
    cid = ctx.define_short(I(sp["p"]), I(sp["a"]), I(sp["b"]))
    g = np.frombuffer(I(sp["g"]["x"]).to_bytes(32, "big") + I(sp["g"]["y"]).to_bytes(32, "big"), np.uint8)
    r = bench.xof("custom:r", n * 32).reshape(n, 32).copy()
    k = bench.xof("custom:k", n * 32).reshape(n, 32).copy()
    pts, inf = ctx.mul_var(cid, r, np.tile(g, (n, 1)))
    dev = torch.device("cuda", 0)
    dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
    dp2 = dp.roll(1, 0).contiguous()
    dxy = torch.zeros(n, 64, dtype=torch.uint8, device=dev)
    dinf = torch.zeros(n, dtype=torch.uint8, device=dev)
    out = {"lib": os.path.basename(os.environ.get("ELLGPU_LIB", "libellgpu.so")), "n": n}
    for name, fn in (("mul_var", lambda: ctx.mul_var_dev(cid, dk, dp, dxy, dinf)),
                     ("mul_add2", lambda: ctx.mul_add2_dev(cid, dk, dp, dk, dp2, dxy, dinf))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name + "_ms"] = round(ms, 3)
        out[name + "_M_per_s"] = round(n / ms / 1e3, 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
