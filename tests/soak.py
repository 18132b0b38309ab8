#!/usr/bin/env python3
"""Randomised differential soak on the GPU box: large seeded batches of every ladder-shaped
entry point on every short curve, ed25519, curve25519 and a user-defined curve against the C port of the reference's algorithm
(oracle/ec_oracle.c, run on host threads), item by item.  Checker use of oracle/ only.

    python tests/soak.py [--seconds 120] [--seed 1]
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import elliptic_amd
from oracle import c_oracle

CURVES = ["secp256k1", "p192", "p224", "p256", "p384", "p521"]


def rnd(seed, n, w):
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def par(fn, n, threads, *arrays):
    """run fn(slice...) over `threads` slices, concatenate the tuple results"""
    cuts = [n * t // threads for t in range(threads + 1)]
    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(lambda t: fn(*[a[cuts[t]:cuts[t + 1]] if a is not None else None for a in arrays]),
                            range(threads)))
    return tuple(np.concatenate([p[j] for p in parts]) for j in range(len(parts[0])))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    ctx = elliptic_amd.Context(0)
    threads = max(1, min(32, len(os.sched_getaffinity(0))))
    t_end = time.time() + a.seconds
    rounds = 0
    checked = 0
    custom = None
    while time.time() < t_end:
        for curve in CURVES:
            B = elliptic_amd.FIELD_BYTES[curve]
            n = {"p384": 3000, "p521": 1200}.get(curve, 8000)
            tag = "soak:%d:%d:%s" % (a.seed, rounds, curve)
            d, k, k2 = rnd(tag + ":d", n, B), rnd(tag + ":k", n, B), rnd(tag + ":k2", n, B)
            if curve == "p521":
                for x in (d, k, k2):
                    x[:, 0] &= 1
            # sprinkle small / structured scalars
            k[::97] = 0
            k[1::97, :B - 1] = 0
            k2[2::97, : B // 2] = 0xFF
            pts, inf = par(lambda dd: c_oracle.mul(curve, dd), n, threads, d)
            assert not inf.any()
            got = ctx.mul_fixed(curve, d)
            assert np.array_equal(got[0], pts) and not got[1].any(), (curve, "mul_fixed")
            want = par(lambda kk, pp: c_oracle.mul(curve, kk, pp), n, threads, k, pts)
            got = ctx.mul_var(curve, k, pts)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (curve, "mul_var")
            # the row layer (csrc/coop.h, coop_mont.h: one item per wave) serves batches of at most
            # 1 365 items on secp256k1 / p256 / p224 / p192: the same items again in slices of that size
            ROW = 1300 if curve in ("secp256k1", "p256", "p224", "p192") else 0
            for lo in (range(0, n, 4 * ROW) if ROW else ()):
                g2 = ctx.mul_var(curve, k[lo:lo + ROW], pts[lo:lo + ROW])
                assert np.array_equal(g2[1], want[1][lo:lo + ROW]) and np.array_equal(g2[0], want[0][lo:lo + ROW]), (curve, "mul_var, row layer", lo)
            want = par(lambda a1, a2, pp: c_oracle.mul_add(curve, a1, None, a2, pp), n, threads, k, k2, pts)
            got = ctx.mul_add2(curve, k, None, k2, pts)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (curve, "mul_add_g")
            for lo in (range(0, n, 4 * ROW) if ROW else ()):
                g2 = ctx.mul_add2(curve, k[lo:lo + ROW], None, k2[lo:lo + ROW], pts[lo:lo + ROW])
                assert np.array_equal(g2[1], want[1][lo:lo + ROW]) and np.array_equal(g2[0], want[0][lo:lo + ROW]), (curve, "mul_add_g, row layer", lo)
            p1 = np.roll(pts, 1, axis=0)
            want = par(lambda a1, q1, a2, pp: c_oracle.mul_add(curve, a1, q1, a2, pp), n, threads, k, p1, k2, pts)
            got = ctx.mul_add2(curve, k, p1, k2, pts)
            assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), (curve, "mul_add2")
            # sign with the device's deterministic nonces, verify on both sides, corrupt, verify again
            NB = elliptic_amd.ORDER_BYTES[curve]
            z = rnd(tag + ":z", n, min(NB, 32) if curve != "p521" else 64)
            r, s, rec, ok = ctx.ecdsa_sign_det(curve, z, d[:, :NB] if NB == B else d)
            assert ok.all(), (curve, "sign_det")
            for lo in (range(0, n, 4 * ROW) if ROW else ()):       # EC#sign's middle on the row layer: the same signatures
                g2 = ctx.ecdsa_sign_det(curve, z[lo:lo + ROW], (d[:, :NB] if NB == B else d)[lo:lo + ROW])
                assert all(np.array_equal(x, y[lo:lo + ROW]) for x, y in zip(g2, (r, s, rec, ok))), (curve, "sign_det, row layer", lo)
            z[::5, 0] ^= 1
            s[1::5, NB - 1] ^= 2
            want = c_oracle.verify(curve, z, r, s, pts, threads=threads)
            got = ctx.ecdsa_verify(curve, z, r, s, pts)
            assert np.array_equal(got, want), (curve, "verify")
            for lo in (range(0, n, 4 * ROW) if ROW else ()):
                g2 = ctx.ecdsa_verify(curve, z[lo:lo + ROW], r[lo:lo + ROW], s[lo:lo + ROW], pts[lo:lo + ROW])
                assert np.array_equal(g2, want[lo:lo + ROW]), (curve, "verify, row layer", lo)
            assert want[2::5].all() and not want[::5].any()
            # keys moved off the curve (the last byte of y flipped: still a field element): exactly
            # those items come back with status 2 -- unless r / s are out of range, which wins --
            # and every other item is untouched
            bad = pts.copy()
            off = np.arange(3, n, 41)
            bad[off, 2 * B - 1] ^= 1
            got2, st2 = ctx.ecdsa_verify(curve, z, r, s, bad, status=True)
            flagged = st2 == 2
            assert got2.max() <= 1 and not got2[flagged].any(), (curve, "verdicts are a mask")
            assert np.array_equal(got2[~flagged], want[~flagged]), (curve, "verify beside off-curve keys")
            assert not flagged[np.setdiff1d(np.arange(n), off)].any() and flagged[off].sum() >= len(off) - 2, (curve, "off-curve flags")
            xy2, inf2 = ctx.mul_var(curve, k, bad)
            assert (inf2[off] == 2).all() and not xy2[off].any(), (curve, "mul_var off-curve")
            on = np.ones(n, bool)
            on[off] = False
            w_xy, w_inf = par(lambda kk, pp: c_oracle.mul(curve, kk, pp), n, threads, k, pts)
            assert np.array_equal(xy2[on], w_xy[on]) and np.array_equal(inf2[on], w_inf[on]), (curve, "mul_var beside off-curve points")
            checked += 8 * n
        # ed25519 (Edwards) and curve25519 (x-only Montgomery): variable base against the C oracle's
        # _extAdd / _extDbl and diffAdd / dbl; a user-defined curve (brainpoolP256r1) against its
        # generic-a _dbl path
        n = 6000
        tag = "soak:%d:%d:25519" % (a.seed, rounds)
        k, r0 = rnd(tag + ":k", n, 32), rnd(tag + ":r", n, 32)
        k[::53] = 0
        k[1::53, :31] = 0
        base = c_oracle.ed_mul(r0, None, threads=threads)
        want = c_oracle.ed_mul(k, base, threads=threads)
        got = ctx.mul_var("ed25519", k, base)
        assert np.array_equal(got[0], want), "ed25519 mul_var"
        gotf = ctx.mul_fixed("ed25519", r0)
        assert np.array_equal(gotf[0], base), "ed25519 mul_fixed"
        xs = rnd(tag + ":x", n, 32)
        xs[:, 0] &= 0x7F
        wx, wi = c_oracle.mont_mul(k, xs, threads=threads)
        gx, gi = ctx.x25519(k, xs)
        assert np.array_equal(gi, wi) and np.array_equal(gx, wx), "x25519"
        if custom is None:
            import parity_checks as PC
            from golden_util import I
            sp = [c for c in PC.custom_curves() if c["name"] == "brainpoolP256r1"][0]
            cid = ctx.define_short(I(sp["p"]), I(sp["a"]), I(sp["b"]))
            cname = c_oracle.define_short("custom:brainpoolP256r1", I(sp["p"]), I(sp["a"]), I(sp["b"]), I(sp["n"]),
                                          I(sp["g"]["x"]), I(sp["g"]["y"]))
            g = np.frombuffer(I(sp["g"]["x"]).to_bytes(32, "big") + I(sp["g"]["y"]).to_bytes(32, "big"), np.uint8)
            custom = (cid, cname, g)
        cid, cname, g = custom
        m = 4000
        cp, ci = ctx.mul_var(cid, r0[:m], np.tile(g, (m, 1)))
        want = c_oracle.mul_mt(cname, k[:m], cp, threads)
        got = ctx.mul_var(cid, k[:m], cp)
        assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]), "custom mul_var"
        checked += 3 * n + m
        rounds += 1
    print(json.dumps({"ok": True, "rounds": rounds, "items_checked": checked, "threads": threads}))


if __name__ == "__main__":
    main()
