#!/usr/bin/env python3
"""One-wave workgroups against two-wave ones per curve (developer A/B; the threshold is read when a
context is created, so the two settings are two contexts):  python tools/wg_curves_ab.py"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import elliptic_amd


def main():
    dev = torch.device("cuda", 0)
    ctxs = {}
    for name, val in (("two-wave groups", "0"), ("default", None)):
        if val is None:
            os.environ.pop("ELLGPU_ONE_WAVE_GROUPS", None)
        else:
            os.environ["ELLGPU_ONE_WAVE_GROUPS"] = val
        ctxs[name] = elliptic_amd.Context(0)
    os.environ.pop("ELLGPU_ONE_WAVE_GROUPS", None)
    rng = np.random.default_rng(5)
    for curve, B in (("p521", 66), ("p384", 48), ("p256", 32), ("p224", 28), ("ed25519", 32), ("secp256k1", 32)):
        for n in (262144, 196608, 131072, 65536, 16384):
            k = rng.integers(0, 256, (n, B), dtype=np.uint8)
            if curve == "p521":
                k[:, 0] &= 1
            if curve == "ed25519":
                k[:, 0] &= 15
            any_ctx = ctxs["default"]
            pts, _ = any_ctx.mul_fixed(curve, k)
            dk, dp = torch.from_numpy(k).to(dev), torch.from_numpy(pts).to(dev)
            xy = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
            inf = torch.zeros(n, dtype=torch.uint8, device=dev)
            row = {"curve": curve, "n": n}
            ref = None
            for name, c in ctxs.items():
                for op, fn in (("mul_var", lambda: c.mul_var_dev(curve, dk, dp, xy, inf)),
                               ("mul_fixed", lambda: c.mul_fixed_dev(curve, dk, xy, inf))):
                    for _ in range(2):
                        fn()
                    torch.cuda.synchronize()
                    reps = 6 if curve in ("p521", "p384") else 12
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    torch.cuda.synchronize()
                    row["%s %s ms" % (op, name)] = round((time.perf_counter() - t0) / reps * 1e3, 4)
                    if op == "mul_var":
                        got = xy.cpu().numpy().copy()
                        if ref is None:
                            ref = got
                        row["same_bytes"] = bool(np.array_equal(ref, got))
            print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
