#!/usr/bin/env python3
"""Driver of tools/microbench/coop_field.hip (GPU box): the lanes-per-item field layer against the
one-item-per-lane field the kernels use, on a LONE wave (blocks = 1: what a single EC#verify /
Point#mul sees) and on grids of 256 / 1024 / 4096 one-wave workgroups.

  python tools/microbench/coop_field.py [out.json]

1. checks the cooperative results exactly (Python integers mod p) -- short chains of products,
   squares, add/sub and Jacobian doublings;
2. times dependent chains of both layers and prints the ratios the round-4 review asked for
   (gate: >= 1.6x per multiplication on an otherwise idle SIMD)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
P = 2 ** 256 - 2 ** 32 - 977
M29 = (1 << 29) - 1


def build():
    exe = os.path.join(HERE, "_build", "coop_field")
    src = os.path.join(HERE, "coop_field.hip")
    if not os.path.exists(exe) or os.path.getmtime(exe) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-o", exe, src], check=True)
    return exe


def limb(s, l):
    """the kernel's seeded operand limbs (k_chain's lambda)"""
    x = (s * 2654435761 + l * 0x9E3779B9) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x85EBCA6B) & 0xFFFFFFFF
    x ^= x >> 13
    return 0 if l > 8 else x & (0xFFFFFF if l == 8 else M29)


def val(limbs):
    return sum(int(v) << (29 * i) for i, v in enumerate(limbs)) % P


def n_form(limbs):
    return all(abs(v) <= (1 << 29) + (1 << 24) for v in limbs[:8]) and -(1 << 5) < limbs[8] < (1 << 25) + (1 << 5)


def dbl(X, Y, Z):
    """short.js:668-737 in the general-Z form the kernel computes (dbl-2009-l)"""
    a, b = X * X % P, Y * Y % P
    c = b * b % P
    d = 2 * ((X + b) ** 2 - a - c) % P
    e = 3 * a % P
    f = e * e % P
    x3 = (f - 2 * d) % P
    y3 = (e * (d - x3) - 8 * c) % P
    z3 = 2 * Y * Z % P
    return x3, y3, z3


def check(res):
    seed = res["seed"]
    x0, y0, z0 = (val([limb(seed + k, l) for l in range(9)]) for k in range(3))
    bad = []
    for n in (1, 2, 3):
        x = x0
        for _ in range(n):
            x = x * y0 % P
        c = res["checks"]["mul%d" % n]
        if val(c["x"]) != x or not n_form(c["x"]) or val(c["y"]) != y0:
            bad.append("mul%d" % n)
        x = x0
        for _ in range(n):
            x = x * x % P
        c = res["checks"]["sqr%d" % n]
        if val(c["x"]) != x or not n_form(c["x"]):
            bad.append("sqr%d" % n)
        x = x0
        for _ in range(n):
            x = (x + y0 - z0) % P
        c = res["checks"]["addsub%d" % n]
        if val(c["x"]) != x or not n_form(c["x"]):
            bad.append("addsub%d" % n)
        pt = (x0, y0, z0)
        for _ in range(n):
            pt = dbl(*pt)
        c = res["checks"]["dbl%d" % n]
        if (val(c["X"]), val(c["Y"]), val(c["Z"])) != pt or not all(n_form(c[k]) for k in "XYZ"):
            bad.append("dbl%d" % n)
    return bad


def main():
    exe = build()
    runs = []
    for seed in (12345, 777, 31337):
        p = subprocess.run([exe, str(seed)], capture_output=True, text=True, check=True)
        res = json.loads(p.stdout)
        res["mismatches"] = check(res)
        runs.append(res)
    out = {"cooperative_checks_failed": sum(len(r["mismatches"]) for r in runs),
           "cooperative_mismatches": [r["mismatches"] for r in runs],
           "cooperative_ns_per_op": runs[0]["ns_per_op"]}
    # the one-item-per-lane layer (the product's own field, through the C ABI's probe)
    import elliptic_amd
    ctx = elliptic_amd.Context(0)
    kinds = {"mul": 10, "sqr": 11, "addsub": 13, "dbl": 14, "madd": 15}
    one = {}
    for blocks in (1, 256, 1024, 4096):
        row = {}
        for name, kind in kinds.items():
            iters = 3000 if kind >= 14 else 20000
            best = None
            for _ in range(3):
                ms, ops = ctx.probe_valu(kind, blocks, iters)
                best = ms if best is None else min(best, ms)
            row[name] = round(best * 1e6 / iters, 2)
        one[str(blocks)] = row
    out["one_lane_ns_per_op"] = one
    out["ratio_one_lane_over_cooperative"] = {
        b: {k: round(one[b][k] / out["cooperative_ns_per_op"][b][k], 2) for k in ("mul", "sqr", "addsub", "dbl")}
        for b in one}
    out["gate"] = {"asked": ">= 1.6x per multiplication on an otherwise idle SIMD (blocks = 1)",
                   "measured_mul": out["ratio_one_lane_over_cooperative"]["1"]["mul"],
                   "measured_sqr": out["ratio_one_lane_over_cooperative"]["1"]["sqr"],
                   "measured_dbl": out["ratio_one_lane_over_cooperative"]["1"]["dbl"]}
    text = json.dumps(out, indent=1)
    print(text)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(text + "\n")
    return 1 if out["cooperative_checks_failed"] else 0


if __name__ == "__main__":
    sys.exit(main())
