#!/usr/bin/env python3
"""Where one call of each single-item operation spends its time: HIP-event time per kernel and the
wall time of the host-buffer call (developer tool, GPU box).

    python tools/single_call_breakdown.py
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
    hz, hr, hs, hq, want = bench.make_signatures(ctx, 64, "latency")
    z, r, s, q = hz[:1].copy(), hr[:1].copy(), hs[:1].copy(), hq[:1].copy()
    edm = [bytes(range(32))]
    eds = np.frombuffer(bytes(range(1, 33)), np.uint8).reshape(1, 32).copy()
    sig, pub = ctx.eddsa_sign(edm, eds)[:2]
    ops = {
        "ecdsa_verify": lambda: ctx.ecdsa_verify("secp256k1", z, r, s, q),
        "mul_var": lambda: ctx.mul_var("secp256k1", r, q),
        "mul_fixed": lambda: ctx.mul_fixed("secp256k1", r),
        "mul_add2 (k1 G + k2 P)": lambda: ctx.mul_add2("secp256k1", r, None, s, q),
        "ecdsa_sign_det": lambda: ctx.ecdsa_sign_det("secp256k1", z, r),
        "ecdsa_recover": lambda: ctx.ecdsa_recover("secp256k1", z, r, s, np.zeros(1, np.uint8)),
        "eddsa_verify": lambda: ctx.eddsa_verify(edm, sig, pub),
        "eddsa_sign": lambda: ctx.eddsa_sign(edm, eds),
        "p256 ecdsa_verify (any tuple)": lambda: ctx.ecdsa_verify("p256", z, r, s, p256q),
        "p256 mul_fixed": lambda: ctx.mul_fixed("p256", r),
    }
    p256q = ctx.mul_fixed("p256", r)[0]
    # the wide NIST curves (round 6: one item per wave on csrc/coop_wide.h)
    for cname, nb in (("p384", 48), ("p521", 66)):
        kk = np.frombuffer(bytes((7 * i + 3) & 0xFF for i in range(nb)), np.uint8).reshape(1, nb).copy()
        kk[0, 0] &= 1
        qq = ctx.mul_fixed(cname, kk)[0]
        ops["%s ecdsa_verify (any tuple)" % cname] = (lambda c=cname, k=kk, q_=qq: ctx.ecdsa_verify(c, k, k, k, q_))
        ops["%s mul_var" % cname] = (lambda c=cname, k=kk, q_=qq: ctx.mul_var(c, k, q_))
        ops["%s mul_fixed" % cname] = (lambda c=cname, k=kk: ctx.mul_fixed(c, k))
        ops["%s ecdsa_sign_det" % cname] = (lambda c=cname, k=kk: ctx.ecdsa_sign_det(c, k, k))
    for name, fn in ops.items():
        for _ in range(3):
            fn()
        ctx.set_timing(True)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        tm = ctx.get_timing()
        ctx.set_timing(False)
        ts.sort()
        print(json.dumps({"op": name, "n": 1, "call_median_us": round(ts[len(ts) // 2] * 1e6, 1), "call_best_us": round(ts[0] * 1e6, 1),
                          "kernels_us": {k: round(v[1] / max(v[0], 1) * 1e3, 1) for k, v in tm.items()},
                          "launches_per_call": round(sum(v[0] for v in tm.values()) / 50, 1)}), flush=True)


if __name__ == "__main__":
    main()
