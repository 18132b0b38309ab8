#!/usr/bin/env python3
"""Secondary throughput table for the other BASELINE.json configs (not the
bench.py headline): fixed-base G*k, variable-base P*k (secp256k1), ed25519 P*k,
p384 P*k, x25519 -- kernel-only, inputs resident in HBM.  GPU box only.

    python tools/bench_configs.py [--reps 3]
"""
import argparse
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import elliptic_amd


def rnd(seed, n, w):
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--curves", default="", help="comma list; default all")
    a = ap.parse_args()
    ctx = elliptic_amd.Context(0)
    dev = torch.device("cuda", 0)
    rows = []

    def timed(name, n, fn):
        fn()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        tm = ctx.get_timing()
        ctx.set_timing(False)
        rows.append({"config": name, "n": n, "items_per_s": n / dt, "ms": dt * 1e3,
                     "kernels_ms": {k: v[1] / v[0] for k, v in tm.items()}})
        print(json.dumps(rows[-1]), flush=True)

    only = [c for c in a.curves.split(",") if c]
    for curve, n in (("secp256k1", 1 << 20), ("p192", 1 << 19), ("p224", 1 << 19), ("p256", 1 << 19), ("p384", 1 << 18), ("p521", 1 << 18),
                     ("ed25519", 1 << 20)):
        if only and curve not in only:
            continue
        B = elliptic_amd.FIELD_BYTES[curve]
        k = rnd("cfg:k:" + curve, n, B)
        if curve == "p521":
            k[:, 0] &= 1
        d = rnd("cfg:d:" + curve, n, B)
        if curve == "p521":
            d[:, 0] &= 1
        dk, dd = torch.from_numpy(k).to(dev), torch.from_numpy(d).to(dev)
        pts = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
        out = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
        inf = torch.zeros(n, dtype=torch.uint8, device=dev)
        timed("%s fixed-base G*k" % curve, n, lambda: ctx.mul_fixed_dev(curve, dd, pts, inf))
        timed("%s variable-base P*k" % curve, n, lambda: ctx.mul_var_dev(curve, dk, pts, out, inf))
        timed("%s k1*G + k2*P" % curve, n, lambda: ctx.mul_add2_dev(curve, dd, None, dk, pts, out, inf))
        if curve in ("secp256k1", "ed25519"):
            # Point#add of the two batches just produced (d*G and k1*G + k2*P), batched normalization
            sm = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
            timed("%s affine point addition P + Q" % curve, n, lambda: ctx.point_add_dev(curve, pts, out, sm, inf))
        if curve in ("secp256k1", "p192", "p224", "p256", "p384", "p521"):
            # ECDSA sign for supplied nonces (hash = k bytes, priv = d, nonce = k ^ d: all < 2^256, a few
            # percent >= n are flagged per item), key decompression of the x coordinates just produced
            NB = elliptic_amd.ORDER_BYTES[curve]
            # p521: 65-byte digests -- recoverPubKey does not truncate e (ec/index.js:231-254), so
            # only digests no longer than n recover the signing key
            hz = dk if curve != "p521" else dk[:, 1:].contiguous()
            nonce = torch.bitwise_xor(dk, dd)
            r_o = torch.zeros((n, NB), dtype=torch.uint8, device=dev)
            s_o = torch.zeros((n, NB), dtype=torch.uint8, device=dev)
            rec = torch.zeros(n, dtype=torch.uint8, device=dev)
            ok = torch.zeros(n, dtype=torch.uint8, device=dev)
            timed("%s ECDSA sign (nonces supplied)" % curve, n,
                  lambda: ctx.ecdsa_sign_dev(curve, hz, dd, nonce, r_o, s_o, rec, ok, canonical=True))
            ok_d = torch.zeros(n, dtype=torch.uint8, device=dev)
            timed("%s ECDSA sign (deterministic nonces, HMAC-DRBG on the device)" % curve, n,
                  lambda: ctx.ecdsa_sign_det_dev(curve, hz, dd, r_o, s_o, rec, ok_d, canonical=True))
            assert bool(ok_d.all())
            ok.copy_(ok_d)
            good = ok.bool()
            # ECDSA verify of those signatures against d*G (the bench.py headline is this row for
            # secp256k1, with its own signature generator)
            ok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
            timed("%s ECDSA verify" % curve, n, lambda: ctx.ecdsa_verify_dev(curve, hz, r_o, s_o, pts, ok2))
            assert bool(ok2[good].bool().all())
            # public-key recovery from those signatures (decompress R, r^-1, s1*G + s2*R): must
            # give back d*G wherever the signing pass accepted the nonce
            q_o = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
            st = torch.zeros(n, dtype=torch.uint8, device=dev)
            timed("%s ECDSA public-key recovery" % curve, n,
                  lambda: ctx.ecdsa_recover_dev(curve, hz, r_o, s_o, rec, q_o, st))
            assert bool((st[good] == 0).all()) and torch.equal(q_o[good], pts[good])
            xs = pts[:, :B].contiguous()
            odd = (pts[:, 2 * B - 1] & 1).contiguous()
            timed("%s key decompression (pointFromX)" % curve, n, lambda: ctx.decompress_dev(curve, xs, odd, out, ok))
            assert torch.equal(out, pts) and bool(ok.all())
            if curve == "secp256k1":
                # SEC1 codecs and KeyPair#validate: encode d*G compressed, decode it back, validate
                # it (curve equation + n*P == O: one more variable-base ladder per key)
                enc = torch.zeros((n, 1 + B), dtype=torch.uint8, device=dev)
                timed("%s encode (compressed SEC1)" % curve, n, lambda: ctx.encode_points_dev(curve, pts, True, enc))
                timed("%s decodePoint (compressed SEC1)" % curve, n, lambda: ctx.decode_points_dev(curve, enc, out, st))
                assert torch.equal(out, pts) and bool((st == 0).all())
                timed("%s KeyPair#validate (curve equation only)" % curve, n,
                      lambda: ctx.validate_dev(curve, pts, None, False, st))
                assert bool((st == 0).all())
                timed("%s KeyPair#validate (with n*P == O)" % curve, n,
                      lambda: ctx.validate_dev(curve, pts, None, True, st))
                assert bool((st == 0).all())
                # EC#verify on wire formats: DER signatures (built on the host from r_o, s_o with
                # the engine's own toDER) + the compressed keys -> decode, parse, verify on device
                ders = ctx.sig_to_der(curve, r_o.cpu().numpy(), s_o.cpu().numpy())
                der_np, len_np = ctx._pack_records(ders)
                der_d = torch.from_numpy(der_np).to(dev)
                len_d = torch.from_numpy(len_np.view(np.int32)).to(dev)
                err = torch.zeros(n, dtype=torch.uint8, device=dev)
                timed("%s ECDSA verify, DER signatures + compressed keys" % curve, n,
                      lambda: ctx.ecdsa_verify_wire_dev(curve, dk, der_d, len_d, enc, ok2, err))
                assert bool(ok2[good].bool().all()) and bool((err == 0).all())
        if curve == "ed25519":
            # EdDSA verify on valid signatures (A = aG, R = rG, S = r + h a; built with the fixed-base
            # kernel + hashlib), 48-byte messages
            N = 2 ** 252 + 27742317777372353535851937790883648493          # the order of the ed25519 base point
            m, mlen = 1 << 18, 48
            raw = rnd("cfg:eddsa", m, 64 + mlen)
            av = [int.from_bytes(raw[i, :32].tobytes(), "little") % N for i in range(m)]
            rv = [int.from_bytes(raw[i, 32:64].tobytes(), "little") % N for i in range(m)]
            msgs = np.ascontiguousarray(raw[:, 64:])
            A, _ = ctx.mul_fixed("ed25519", elliptic_amd.ints_to_be(av, 32))
            R, _ = ctx.mul_fixed("ed25519", elliptic_amd.ints_to_be(rv, 32))

            def enc(P):
                y = P[:, 32:][:, ::-1].copy()
                y[:, 31] |= ((P[:, 31] & 1) << 7).astype(np.uint8)
                return y
            Ae, Re = enc(A), enc(R)
            sig = np.zeros((m, 64), np.uint8)
            sig[:, :32] = Re
            for i in range(m):
                h = int.from_bytes(hashlib.sha512(Re[i].tobytes() + Ae[i].tobytes() + msgs[i].tobytes()).digest(),
                                   "little") % N
                sig[i, 32:] = np.frombuffer(((rv[i] + h * av[i]) % N).to_bytes(32, "little"), np.uint8)
            # the 2^18 distinct signatures tiled four times: 2^20 items, as in the other rows
            dm, ds, dp = [torch.from_numpy(x).to(dev).repeat(4, 1).contiguous() for x in (msgs, sig, Ae)]
            ok = torch.zeros(4 * m, dtype=torch.uint8, device=dev)
            timed("ed25519 EdDSA verify (48-byte messages)", 4 * m, lambda: ctx.eddsa_verify_dev(dm, mlen, ds, dp, ok))
            assert bool(ok.all())
            # EdDSA sign from 32-byte secrets (two hashes of the message, a*G, r*G, S), then verified
            sec = torch.from_numpy(rnd("cfg:eddsa:secret", 4 * m, 32)).to(dev)
            sg = torch.zeros((4 * m, 64), dtype=torch.uint8, device=dev)
            pk = torch.zeros((4 * m, 32), dtype=torch.uint8, device=dev)
            timed("ed25519 EdDSA sign (48-byte messages)", 4 * m, lambda: ctx.eddsa_sign_dev(sec, dm, mlen, sg, pk))
            ctx.eddsa_verify_dev(dm, mlen, sg, pk, ok)
            torch.cuda.synchronize()
            assert bool(ok.all())
    if only and "curve25519" not in only:
        return
    n = 1 << 20
    k = torch.from_numpy(rnd("cfg:k:x", n, 32)).to(dev)
    x = torch.from_numpy(rnd("cfg:x:x", n, 32)).to(dev)
    ox = torch.zeros((n, 32), dtype=torch.uint8, device=dev)
    inf = torch.zeros(n, dtype=torch.uint8, device=dev)
    timed("curve25519 x-only ladder", n, lambda: ctx.x25519_dev(k, x, ox, inf))


if __name__ == "__main__":
    main()
