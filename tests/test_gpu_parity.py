"""GPU parity tests proper (-m gpu): libellgpu.so's HIP kernels, called through
the C ABI, against (1) the reference's golden vectors, (2) the oracle on seeded
random inputs, (3) size-independent properties at BASELINE.json's full sizes."""
import hashlib
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import elliptic_amd  # noqa: E402
from elliptic_amd import be_to_ints, ints_to_be  # noqa: E402
import parity_checks as PC  # noqa: E402
from oracle import ec_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = elliptic_amd.Context(0)          # raises if libellgpu.so or the GPU is missing
    yield c
    c.close()


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_mul_golden(ctx, curve):
    assert PC.check_mul_golden(ctx, curve) > 50


@pytest.mark.parametrize("curve", ["secp256k1", "p192", "p256", "p384", "p521"])
def test_exceptional_keys_and_scalars(ctx, curve):
    """keys +-G, 2G, +-lambda G ... with (u1, u2) that send u1 G + u2 Q through O, +-G, 2G, Q and GLV
    halves of 0 / 1: k1 G + k2 Q, k1 P1 + k2 P2 and the verdicts against the C oracle"""
    assert PC.check_exceptional_keys(ctx, curve) > 400


def test_x25519_golden(ctx):
    assert PC.check_x25519_golden(ctx) > 30


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_offcurve_operands_are_reported_not_guessed(ctx, curve):
    """offcurve_<curve>.json: points that are not on the curve get status 2 (out_inf / out_ok),
    a zeroed result, and leave the on-curve items of the same batch untouched"""
    assert PC.check_offcurve_golden(ctx, curve) >= 29


def test_offcurve_keys_scattered_in_a_full_size_batch(ctx, monkeypatch):
    """2^20 verifies with every 1000th key moved off the curve (y + 1): exactly those items -- minus
    the ones whose r / s are out of range, which the reference rejects before it touches the key
    -- have status 2 beside a verdict of 0, every other verdict is the expected mask; same for the small-grid tuning on a
    131 072-item shard and for P*k through the host-buffer pipeline."""
    n = 1 << 20
    h, r, s, pub, expect = _make_sigs(ctx, n, "gpu-test-fullsize")
    pub = pub.copy()
    off = np.arange(7, n, 1000)
    p = O.get_curve("secp256k1").p
    for i in off:
        y = (int.from_bytes(pub[i, 32:].tobytes(), "big") + 1) % p
        pub[i, 32:] = np.frombuffer(y.to_bytes(32, "big"), np.uint8)
    want = expect.copy()
    want[off] = 0                        # the verdicts are a mask: 0 for a key that is no curve point
    wst = np.zeros(n, np.uint8)
    wst[off] = 2
    nn = O.get_curve("secp256k1").n
    for i in off:                        # corrupted r / s can leave [1, n): rejected first
        ri, si = int.from_bytes(r[i].tobytes(), "big"), int.from_bytes(s[i].tobytes(), "big")
        if not (0 < ri < nn and 0 < si < nn):
            wst[i] = 0
    got, gst = ctx.ecdsa_verify("secp256k1", h, r, s, pub, status=True)
    assert np.array_equal(got, want) and np.array_equal(gst, wst)
    assert np.array_equal(ctx.ecdsa_verify("secp256k1", h, r, s, pub), want)      # without the status array
    m = 131072
    got, gst = ctx.ecdsa_verify("secp256k1", h[:m], r[:m], s[:m], pub[:m], status=True)
    assert np.array_equal(got, want[:m]) and np.array_equal(gst, wst[:m])
    monkeypatch.setenv("ELLGPU_SMALL_GRID", "0")            # the full-grid tuning on the same shard
    c2 = elliptic_amd.Context(0)                            # (tuning overrides are read at creation)
    monkeypatch.delenv("ELLGPU_SMALL_GRID")
    got, gst = c2.ecdsa_verify("secp256k1", h[:m], r[:m], s[:m], pub[:m], status=True)
    assert np.array_equal(got, want[:m]) and np.array_equal(gst, wst[:m])
    c2.close()
    # device-resident form: the status array is optional there too
    import torch
    dv = [torch.from_numpy(x[:m]).cuda() for x in (h, r, s, pub)]
    dok = torch.full((m,), 9, dtype=torch.uint8, device="cuda")
    dst = torch.full((m,), 9, dtype=torch.uint8, device="cuda")
    ctx.ecdsa_verify_dev("secp256k1", *dv, dok, out_status=dst)
    ctx.synchronize()
    assert np.array_equal(dok.cpu().numpy(), want[:m]) and np.array_equal(dst.cpu().numpy(), wst[:m])
    m = 300000
    xy, inf = ctx.mul_var("secp256k1", r[:m], pub[:m])
    is_off = np.zeros(m, bool)
    is_off[off[off < m]] = True
    assert (inf[is_off] == 2).all() and not xy[is_off].any() and (inf[~is_off] == 0).all()
    xy0, inf0 = ctx.mul_var("secp256k1", r[:2048], pub[:2048])
    from oracle import c_oracle
    wxy, winf = c_oracle.mul("secp256k1", r[:2048], pub[:2048])    # the oracle computes off-curve items too
    on = ~is_off[:2048]
    assert np.array_equal(xy0[on], wxy[on]) and np.array_equal(inf0[on], winf[on])


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_verify_golden(ctx, curve):
    assert PC.check_verify_golden(ctx, curve) > 15


@pytest.mark.parametrize("curve", ["secp256k1", "p192", "p256", "p384", "p521", "ed25519"])
def test_decompress_golden(ctx, curve):
    assert PC.check_decompress_golden(ctx, curve) > 40


def test_decompress_roundtrip_full_size(ctx):
    """2^18 public keys: decompress(x, parity(y)) must give back y exactly, and feeding the
    decompressed keys to the verifier must give the same mask as the original keys."""
    n = 1 << 18
    h, r, s, pub, expect = _make_sigs(ctx, n, "gpu-test-decompress")
    odd = (pub[:, 63] & 1).astype(np.uint8)
    out, ok = ctx.decompress("secp256k1", np.ascontiguousarray(pub[:, :32]), odd)
    assert ok.all() and np.array_equal(out, pub)
    out2, ok2 = ctx.decompress("secp256k1", np.ascontiguousarray(pub[:, :32]), 1 - odd)
    assert ok2.all() and np.array_equal(out2[:, :32], pub[:, :32]) and not np.array_equal(out2[:, 32:], pub[:, 32:])
    assert np.array_equal(ctx.ecdsa_verify("secp256k1", h, r, s, out), expect)


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_sign_deterministic_golden(ctx, curve):
    assert PC.check_signdet_golden(ctx, curve) >= 10


@pytest.mark.parametrize("curve", ["secp256k1", "p192", "p256", "p384", "p521"])
def test_recover_golden(ctx, curve):
    assert PC.check_recover_golden(ctx, curve) >= 30


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_codec_golden(ctx, curve):
    assert PC.check_codec_golden(ctx, curve) >= 100


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_wire_golden(ctx, curve):
    assert PC.check_wire_golden(ctx, curve) >= 60


@pytest.mark.parametrize("curve", ["secp256k1", "p256", "p521"])
def test_ecdh_derive(ctx, curve):
    assert PC.check_ecdh(ctx, curve) > 0


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_point_add_golden(ctx, curve):
    assert PC.check_add_golden(ctx, curve) >= 20


@pytest.mark.parametrize("curve", ["secp256k1", "p256", "p384", "p521"])
def test_codec_random_vs_oracle(ctx, curve):
    assert PC.check_codec_random(ctx, curve, n=3000) > 0


def test_p521_paired_wave_kernels(ctx):
    """p521 batches above one single-wave round run a second instantiation of the ladder kernels
    (two waves per SIMD): same results as the one-wave kernels the smaller batches use"""
    n, B = 74 * 1024, 66
    rng = np.random.default_rng(521)
    k = np.frombuffer(rng.bytes(n * B), np.uint8).reshape(n, B).copy()
    d = np.frombuffer(rng.bytes(n * B), np.uint8).reshape(n, B).copy()
    k[:, 0] &= 1
    d[:, 0] &= 1
    h = n // 2
    pts, _ = ctx.mul_fixed("p521", d)
    assert np.array_equal(pts, np.concatenate([ctx.mul_fixed("p521", d[:h])[0], ctx.mul_fixed("p521", d[h:])[0]]))
    big, binf = ctx.mul_var("p521", k, pts)
    lo, linf = ctx.mul_var("p521", k[:h], pts[:h])
    hi, hinf = ctx.mul_var("p521", k[h:], pts[h:])
    assert np.array_equal(big, np.concatenate([lo, hi])) and np.array_equal(binf, np.concatenate([linf, hinf]))
    big, binf = ctx.mul_add2("p521", d, None, k, pts)
    lo, linf = ctx.mul_add2("p521", d[:h], None, k[:h], pts[:h])
    hi, hinf = ctx.mul_add2("p521", d[h:], None, k[h:], pts[h:])
    assert np.array_equal(big, np.concatenate([lo, hi])) and np.array_equal(binf, np.concatenate([linf, hinf]))
    z = np.ascontiguousarray(k[:, 1:])
    r, s_, rec, ok = ctx.ecdsa_sign_det("p521", z, d)
    assert ok.all()
    r[::7, 40] ^= 1                                            # some must fail
    v = ctx.ecdsa_verify("p521", z, r, s_, pts)
    v2 = np.concatenate([ctx.ecdsa_verify("p521", z[:h], r[:h], s_[:h], pts[:h]),
                         ctx.ecdsa_verify("p521", z[h:], r[h:], s_[h:], pts[h:])])
    assert np.array_equal(v, v2) and not v[::7].any() and v[1::7].all()


def test_der_fuzz(ctx):
    assert PC.check_der_fuzz(ctx) > 2000


def test_recover_p224_tonelli_shanks(ctx):
    """p224 (p = 1 mod 4) recovers through the device's Tonelli-Shanks square root"""
    assert PC.check_recover_golden(ctx, "p224") > 20


def test_eddsa_sign_golden(ctx):
    assert PC.check_eddsa_sign_golden(ctx) > 100


def test_eddsa_verify_golden(ctx):
    assert PC.check_eddsa_golden(ctx) > 200


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_sign_golden(ctx, curve):
    assert PC.check_sign_golden(ctx, curve) >= 12


def test_sign_then_verify_full_size(ctx):
    """2^18 signatures made by the engine (random d, k, z) must all verify with the engine's
    verifier against the matching public keys, low-s form respected; a subset vs the oracle."""
    n = 1 << 18
    cur = O.get_curve("secp256k1")
    raw = np.frombuffer(hashlib.shake_256(b"sign-fullsize").digest(n * 96), dtype=np.uint8).reshape(n, 96)
    z, d, k = (np.ascontiguousarray(raw[:, a:a + 32]) for a in (0, 32, 64))
    d[:, 0] &= 0x7F
    k[:, 0] &= 0x7F                                   # < n for sure
    r, s, rec, ok = ctx.ecdsa_sign("secp256k1", z, d, k, canonical=True)
    assert ok.all()
    pub, inf = ctx.mul_fixed("secp256k1", d)
    assert np.array_equal(ctx.ecdsa_verify("secp256k1", z, r, s, pub), np.ones(n, np.uint8))
    half = cur.n >> 1
    rs = random.Random(3)
    for i in [0, 1, 2] + [rs.randrange(n) for _ in range(200)]:
        want = O.ecdsa_sign(cur, int.from_bytes(z[i].tobytes(), "big"), 32, int.from_bytes(d[i].tobytes(), "big"),
                            k[i].tobytes(), canonical=True)
        got = (int.from_bytes(r[i].tobytes(), "big"), int.from_bytes(s[i].tobytes(), "big"), int(rec[i]))
        assert got == want and got[1] <= half


def test_eddsa_full_size_mask(ctx):
    """2^18 synthetic ed25519 signatures (A = aG, R = rG, S = r + h*a, built with the engine's
    own fixed-base kernel and hashlib), every 100th corrupted: exact mask, no decode errors;
    a seeded subset re-checked with the oracle."""
    import torch
    cur = O.get_curve("ed25519")
    n, mlen = 1 << 18, 48
    rnd = np.frombuffer(hashlib.shake_256(b"eddsa-fullsize").digest(n * (64 + mlen)), dtype=np.uint8)
    a = [int.from_bytes(rnd[i * 32:(i + 1) * 32].tobytes(), "little") % cur.n for i in range(n)]
    rr = [int.from_bytes(rnd[(n + i) * 32:(n + i + 1) * 32].tobytes(), "little") % cur.n for i in range(n)]
    msgs = rnd[2 * n * 32:].reshape(n, mlen).copy()
    A, _ = ctx.mul_fixed("ed25519", ints_to_be(a, 32))
    R, _ = ctx.mul_fixed("ed25519", ints_to_be(rr, 32))

    def enc(P):
        y = P[:, 32:][:, ::-1].copy()
        y[:, 31] |= ((P[:, 31] & 1) << 7).astype(np.uint8)
        return y
    Ae, Re = enc(A), enc(R)
    sig = np.zeros((n, 64), np.uint8)
    sig[:, :32] = Re
    for i in range(n):
        h = int.from_bytes(hashlib.sha512(Re[i].tobytes() + Ae[i].tobytes() + msgs[i].tobytes()).digest(), "little") % cur.n
        sig[i, 32:] = np.frombuffer(((rr[i] + h * a[i]) % cur.n).to_bytes(32, "little"), np.uint8)
    expect = np.ones(n, np.uint8)
    for j in range(0, n, 100):
        sig[j, 33 + (j % 20)] ^= 1 << (j % 7)
        expect[j] = 0
    dev = torch.device("cuda", 0)
    dm, ds, dp = [torch.from_numpy(x).to(dev) for x in (msgs, sig, Ae)]
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    err = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.eddsa_verify_dev(dm, mlen, ds, dp, ok, err)
    torch.cuda.synchronize()
    assert np.array_equal(ok.cpu().numpy(), expect) and not err.any()
    rs = random.Random(5)
    for i in [0, 100, 200] + [rs.randrange(n) for _ in range(60)]:
        assert O.eddsa_verify(cur, msgs[i].tobytes(), sig[i].tobytes(), Ae[i].tobytes()) == bool(expect[i])


def _xy(arr, i, B):
    return (int.from_bytes(arr[i, :B].tobytes(), "big"), int.from_bytes(arr[i, B:].tobytes(), "big"))


@pytest.mark.parametrize("curve,count", [("secp256k1", 600), ("p256", 150), ("p384", 60), ("p521", 24),
                                         ("p224", 80), ("p192", 80)])
def test_random_vs_oracle(ctx, curve, count):
    """seeded random scalars/points: fixed, variable, mulAdd against the oracle"""
    cur = O.get_curve(curve)
    B = elliptic_amd.FIELD_BYTES[curve]
    rnd = random.Random("gpu-parity:" + curve)
    ds = [rnd.randrange(1, cur.n) for _ in range(count)]
    ks = [rnd.randrange(0, 1 << (8 * B if curve != "p521" else 521)) for _ in range(count)]
    pub, inf = ctx.mul_fixed(curve, ints_to_be(ds, B))
    assert not inf.any()
    for i in range(0, count, 7):
        w = cur.g.mul(ds[i])
        assert _xy(pub, i, B) == (w.x, w.y)
    out, inf = ctx.mul_var(curve, ints_to_be(ks, B), pub)
    for i in range(count):
        w = cur.point(*_xy(pub, i, B)).mul(ks[i])
        got = None if inf[i] else _xy(out, i, B)
        assert got == (None if w.inf else (w.x, w.y)), (curve, i)
    m = count // 3
    k2 = [rnd.randrange(0, cur.n) for _ in range(m)]
    out, inf = ctx.mul_add2(curve, ints_to_be(ks[:m], B), None, ints_to_be(k2, B), pub[:m])
    out2, inf2 = ctx.mul_add2(curve, ints_to_be(ks[:m], B), np.tile(ints_to_be([cur.g.x, cur.g.y], B).reshape(1, -1), (m, 1)),
                              ints_to_be(k2, B), pub[:m])
    assert np.array_equal(out, out2) and np.array_equal(inf, inf2)
    for i in range(0, m, 3):
        w = cur.g.mul_add(ks[i], cur.point(*_xy(pub, i, B)), k2[i])
        got = None if inf[i] else _xy(out, i, B)
        assert got == (None if w.inf else (w.x, w.y)), (curve, i)


def test_ed25519_and_x25519_random_vs_oracle(ctx):
    cur = O.get_curve("ed25519")
    rnd = random.Random("gpu-parity:ed")
    n = 120
    ds = [rnd.randrange(1, cur.n) for _ in range(n)]
    ks = [rnd.randrange(0, 1 << 256) for _ in range(n)]
    pub, inf = ctx.mul_fixed("ed25519", ints_to_be(ds, 32))
    out, inf = ctx.mul_var("ed25519", ints_to_be(ks, 32), pub)
    for i in range(n):
        R = cur.point(*_xy(pub, i, 32)).mul(ks[i])
        assert _xy(out, i, 32) == R.normalized()
        assert bool(inf[i]) == R.is_infinity()
    mc = O.get_curve("curve25519")
    xs = [rnd.randrange(1, mc.p) for _ in range(n)]
    out, inf = ctx.x25519(ints_to_be(ks, 32), ints_to_be(xs, 32))
    for i in range(n):
        w = mc.mul_x(xs[i], ks[i])
        got = None if inf[i] else int.from_bytes(out[i].tobytes(), "big")
        assert got == w


def test_ragged_and_empty_batches(ctx):
    cur = O.get_curve("secp256k1")
    for n in (0, 1, 63, 64, 65, 127, 129, 1000):
        ks = ints_to_be([(i * 0x9E3779B97F4A7C15 + 1) % cur.n for i in range(n)], 32)
        out, inf = ctx.mul_fixed("secp256k1", ks)
        assert out.shape == (n, 64)
        for i in (0, n - 1) if n else ():
            w = cur.g.mul(int.from_bytes(ks[i].tobytes(), "big"))
            assert _xy(out, i, 32) == (w.x, w.y)


def test_empty_batches_newer_entry_points(ctx):
    z = np.zeros((0, 32), np.uint8)
    e32 = np.zeros((0, 32), np.uint8)
    assert ctx.ecdsa_sign("secp256k1", z, e32, e32)[0].shape == (0, 32)
    assert ctx.ecdsa_sign_det("secp256k1", z, e32)[0].shape == (0, 32)
    assert ctx.ecdsa_recover("secp256k1", z, e32, e32, np.zeros(0, np.uint8))[0].shape == (0, 64)
    assert ctx.decompress("secp256k1", e32, np.zeros(0, np.uint8))[0].shape == (0, 64)
    assert ctx.eddsa_sign([], np.zeros((0, 32), np.uint8))[0].shape == (0, 64)
    assert ctx.eddsa_verify([], np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8))[0].shape == (0,)
    assert ctx.decode_points("secp256k1", np.zeros((0, 33), np.uint8))[0].shape == (0, 64)
    assert ctx.encode_points("secp256k1", np.zeros((0, 64), np.uint8), compact=True).shape == (0, 33)
    assert ctx.validate("ed25519", np.zeros((0, 64), np.uint8)).shape == (0,)
    assert ctx.sig_from_der("p256", [])[0].shape == (0, 32)
    assert ctx.sig_to_der("p256", e32, e32) == []
    assert ctx.ecdsa_verify_wire("secp256k1", z, [], np.zeros((0, 33), np.uint8))[0].shape == (0,)
    assert ctx.point_add("p384", np.zeros((0, 96), np.uint8), np.zeros((0, 96), np.uint8))[0].shape == (0, 96)
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.decode_points("ed25519", np.zeros((1, 33), np.uint8))
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.point_add("curve25519", np.zeros((1, 64), np.uint8), np.zeros((1, 64), np.uint8))


def _make_sigs(ctx, n, seed):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.make_signatures(ctx, n, seed)


def test_full_size_verify_mask_and_device_api(ctx):
    """BASELINE configs[2] at full size (2^20 tuples, 1% corrupted): the ok-mask
    must equal the construction's expected mask exactly; host and device entry
    points must agree; a seeded subset is re-checked with the oracle."""
    import torch
    n = 1 << 20
    h, r, s, pub, expect = _make_sigs(ctx, n, "gpu-test-fullsize")
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x).to(dev) for x in (h, r, s, pub)]
    ok = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.ecdsa_verify_dev("secp256k1", t[0], t[1], t[2], t[3], ok)
    torch.cuda.synchronize()
    got = ok.cpu().numpy()
    assert np.array_equal(got, expect)
    assert int(expect.sum()) == n - (n + 99) // 100
    # host-buffer entry point: pipelined in chunks over two lanes + a copy stream; cut sizes
    # around the residency quantum (196608 items) must not change the mask
    for m in (5000, 196608 + 7, 300000, n):
        assert np.array_equal(ctx.ecdsa_verify("secp256k1", h[:m], r[:m], s[:m], pub[:m]), expect[:m]), m
    m = 700001
    xy_h, inf_h = ctx.mul_var("secp256k1", r[:m], pub[:m])
    xy_d = torch.zeros((m, 64), dtype=torch.uint8, device=dev)
    inf_d = torch.zeros(m, dtype=torch.uint8, device=dev)
    ctx.mul_var_dev("secp256k1", t[1][:m], t[3][:m], xy_d, inf_d)
    torch.cuda.synchronize()
    assert np.array_equal(xy_h, xy_d.cpu().numpy()) and np.array_equal(inf_h, inf_d.cpu().numpy())
    cur = O.get_curve("secp256k1")
    rnd = random.Random(77)
    for i in [0, 100, 200, 300] + [rnd.randrange(n) for _ in range(300)]:
        want = O.ecdsa_verify(cur, int.from_bytes(h[i].tobytes(), "big"), 32,
                              int.from_bytes(r[i].tobytes(), "big"), int.from_bytes(s[i].tobytes(), "big"),
                              cur.point(*_xy(pub, i, 32)))
        assert want == bool(expect[i])


def test_parted_verify_and_mul_small_batches(monkeypatch):
    """Batches that leave most SIMDs idle take the PARTED verify by default (three lanes per item:
    the two GLV half ladders and the comb in different waves, then ecdsa_join -- engine.h
    FnEcdsaParts); ELLGPU_PARTED_GRID=0 keeps them on the one-lane ladder.  Same verdicts from
    both, equal to the construction's mask and (sampled) to the C port of the reference's
    algorithm; off-curve keys, r / s out of range and the group law's exceptional cases included
    (every wave of a part region full, half full, and a lone lane).
    The parts run in two forms: one item per lane (work.h) and, for batches of at most
    ELLGPU_COOP_GRID items (default: two waves per SIMD's worth), one item per WAVE with its field
    elements spread over a 16-lane row (coop.h: DPP / readlane arithmetic, 2.2 x fewer instructions
    on the critical path).  Every form must give the same bytes."""
    from oracle import c_oracle
    monkeypatch.setenv("ELLGPU_PARTED_GRID", "0")
    c0 = elliptic_amd.Context(0)                       # the whole ladder on one lane
    monkeypatch.setenv("ELLGPU_PARTED_GRID", str(1 << 30))
    monkeypatch.setenv("ELLGPU_COOP_GRID", "0")
    monkeypatch.setenv("ELLGPU_ROW_GRID", "0")
    c1 = elliptic_amd.Context(0)                       # parts, one item per lane
    monkeypatch.setenv("ELLGPU_ROW_GRID", str(1 << 30))
    monkeypatch.setenv("ELLGPU_ROW_FROM", "0")
    c3 = elliptic_amd.Context(0)                       # parts, one item per ROW of the wave (four per wave; round 6), whatever the batch
    monkeypatch.setenv("ELLGPU_COOP_GRID", str(1 << 30))
    monkeypatch.setenv("ELLGPU_ROW_FROM", str(1 << 30))
    c2 = elliptic_amd.Context(0)                       # parts, one item per wave (the row layer), whatever the batch
    monkeypatch.delenv("ELLGPU_PARTED_GRID")
    monkeypatch.delenv("ELLGPU_COOP_GRID")
    monkeypatch.delenv("ELLGPU_ROW_GRID")
    monkeypatch.delenv("ELLGPU_ROW_FROM")
    cd = elliptic_amd.Context(0)                       # the default thresholds
    coop_default = 256 * 4 * 4 // 3                    # engine.h Tuning::coop_grid on 256 CUs
    row_from, row_default = 256 * 5 // 2, 256 * 18     # ... Tuning::row_from / row_grid: the one-item-per-row window
    SUFFIXES = ("", "_c", "_r")

    def forms(m):
        """(context, kernel of the parts or None) for a batch of m items"""
        return ((c0, None), (c1, ""), (c2, "_c"), (c3, "_r"),
                (cd, "_r" if row_from < m <= row_default else ("_c" if m <= coop_default else "")))

    def row(c, m):
        """do the calls that have no parted one-lane form (fixed base, decompression, the front of
        the recovery) run on the row layer?  ELLGPU_COOP_GRID alone decides"""
        return c is c2 or (c is not c1 and c is not c3 and m <= coop_default)
    n = 32768
    h, r, s, pub, expect = _make_sigs(c0, n, "gpu-test-parted")
    pub = pub.copy()
    p = O.get_curve("secp256k1").p
    want = expect.copy()
    for i in range(5, n, 97):                          # keys off the curve: status 2
        y = (int.from_bytes(pub[i, 32:].tobytes(), "big") + 1) % p
        pub[i, 32:] = np.frombuffer(y.to_bytes(32, "big"), np.uint8)
        want[i] = 2
    r = r.copy()
    r[11] = 0                                          # r = 0: rejected before the key is looked at
    want[11] = 0
    for m in (1, 2, 3, 63, 64, 65, 127, 128, 129, 640, 641, 1000, 1365, 1366, 1367, 4096, 4608, 4609, 21845, 32768):
        sl = (h[:m], r[:m], s[:m], pub[:m])
        for c, parts in forms(m):
            if c is c2 and m > 4096:
                continue                               # (a wave per part: 3 m workgroups -- small batches are its job)
            c.set_timing(True)
            got, gst = c.ecdsa_verify("secp256k1", *sl, status=True)
            tm = c.get_timing()
            c.set_timing(False)
            # (`want` marks off-curve keys with 2: verdict 0 and status 2 at the C ABI)
            assert got.max(initial=0) <= 1 and not got[gst == 2].any(), (m, parts)
            assert np.array_equal(np.where(gst == 2, 2, got), want[:m]), (m, parts)
            assert ("ecdsa_join" in tm) == (parts is not None), (m, parts, sorted(tm))
            for suffix in SUFFIXES:
                assert ("ecdsa_parts" + suffix in tm) == (parts == suffix), (m, parts, sorted(tm))
    # above the default threshold: the one-lane small-grid ladder
    h2, r2, s2, pub2, expect2 = _make_sigs(c0, 40000, "gpu-test-parted-2")
    cd.set_timing(True)
    assert np.array_equal(cd.ecdsa_verify("secp256k1", h2, r2, s2, pub2), expect2)
    assert "ecdsa_parts" not in cd.get_timing()
    cd.set_timing(False)
    assert np.array_equal(c1.ecdsa_verify("secp256k1", h2, r2, s2, pub2), expect2)
    j = np.arange(0, n, 11)
    on = want[j] != 2
    assert np.array_equal(np.asarray(c_oracle.verify("secp256k1", h[j], r[j], s[j], pub[j]))[on], want[j][on])
    # Point#mul in the parted form (two lanes per item + mul_join): same bytes as the one-lane
    # ladder, the C port's on a sample; off-curve points -> 2; k = 0, n, n + 1, 2^256 - 1 (the
    # halves cancel / wrap) included
    nn = O.get_curve("secp256k1").n
    ks = r.copy()
    for i, kv in enumerate((0, nn, nn + 1, (1 << 256) - 1, 1, 2, nn - 1)):
        ks[20 + i] = np.frombuffer(kv.to_bytes(32, "big"), np.uint8)
    for m in (1, 27, 64, 65, 129, 1366, 4096, 32768):
        outs = []
        for c, parts in forms(m):
            if c is c2 and m > 4096:
                continue
            c.set_timing(True)
            outs.append(c.mul_var("secp256k1", ks[:m], pub[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("mul_join" in tm) == (parts is not None), (m, parts, sorted(tm))
            for suffix in SUFFIXES:
                assert ("mul_parts" + suffix in tm) == (parts == suffix), (m, parts, sorted(tm))
        for xy, inf in outs[1:]:
            assert np.array_equal(xy, outs[0][0]) and np.array_equal(inf, outs[0][1]), m
    xy, inf = outs[1]
    assert (inf[want[:m] == 2] == 2).all() and int((inf == 2).sum()) == int((want[:m] == 2).sum())
    # Point#mul on G (KeyPair#getPublic): a handful of items run the comb and the item's own inversion
    # on a wave (mul_fixed_c); same bytes as mul_fixed -> normalize, k = 0, n, n + 1, 2^256 - 1 included
    for m in (1, 27, 64, 65, 1365, 1366, 4096):
        outs_f = []
        for c, parts in forms(m):
            if c is c2 and m > 4096:
                continue
            c.set_timing(True)
            outs_f.append(c.mul_fixed("secp256k1", ks[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("mul_fixed_c" in tm) == row(c, m) and ("mul_fixed" in tm) != row(c, m), (m, parts, sorted(tm))
        for fxy, finf in outs_f[1:]:
            assert np.array_equal(fxy, outs_f[0][0]) and np.array_equal(finf, outs_f[0][1]), m
    jf = np.concatenate([np.arange(20, 27), np.arange(0, m, 41)])
    wxy, winf = c_oracle.mul("secp256k1", ks[jf])
    assert np.array_equal(outs_f[0][0][jf], wxy) and np.array_equal(outs_f[0][1][jf], winf)
    j = np.concatenate([np.arange(20, 27), np.arange(0, m, 23)])
    j = j[want[j] != 2]
    wxy, winf = c_oracle.mul("secp256k1", ks[j], pub[j])
    assert np.array_equal(xy[j], wxy) and np.array_equal(inf[j], winf)
    # k1 G + k2 P (Point#mulAdd with G, EC#recoverPubKey) in three waves: the same bytes as the one
    # ladder with the comb behind it
    for m in (1, 65, 4096):
        outs = []
        for c, parts in ((c0, None), (c1, ""), (c2, "_c"), (c3, "_r")):
            c.set_timing(True)
            outs.append(c.mul_add2("secp256k1", s[:m], None, ks[:m], pub[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("mul_add_g" in tm) == (parts is None), (m, parts, sorted(tm))
            for suffix in SUFFIXES:
                assert ("mul_parts" + suffix in tm) == (parts == suffix), (m, parts, sorted(tm))
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), m
    j = np.arange(0, m, 37)
    j = j[want[j] != 2]
    wxy, winf = c_oracle.mul_add("secp256k1", s[j], None, ks[j], pub[j])
    assert np.array_equal(outs[1][0][j], wxy) and np.array_equal(outs[1][1][j], winf)
    # ShortCurve#pointFromX and EC#recoverPubKey of a handful of items: the square root on a wave per
    # item (decompress_c; recover_parts_c runs it beside r^-1 in one launch) -- same bytes as the
    # one-lane kernels on abscissas with and without a point, x >= p, every recovery id, r = 0, r >= n
    xs_ = pub[:, :32].copy()
    xs_[1::3, 31] ^= 0x55                              # mostly no points any more
    xs_[7] = 0xFF                                      # x >= p
    odd_ = (np.arange(n) % 2).astype(np.uint8)
    rec_ = (np.arange(n) % 5).astype(np.uint8)         # 4: more than two bits
    rr_ = r.copy()
    rr_[13] = 0xFF                                     # r >= n
    for m in (1, 3, 64, 65, 1365, 1366, 4096):
        outs_d, outs_r = [], []
        for c, parts in forms(m):
            if c is c2 and m > 4096:
                continue
            c.set_timing(True)
            outs_d.append(c.decompress("secp256k1", xs_[:m], odd_[:m]))
            tm = c.get_timing()
            assert ("decompress_c" in tm) == row(c, m) and ("decompress" in tm) != row(c, m), (m, parts, sorted(tm))
            c.set_timing(True)
            outs_r.append(c.ecdsa_recover("secp256k1", h[:m], rr_[:m], s[:m], rec_[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("recover_parts_c" in tm) == row(c, m) and ("recover_prep" in tm) != row(c, m), (m, parts, sorted(tm))
        for o in outs_d[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(outs_d[0], o)), ("decompress", m)
        for o in outs_r[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(outs_r[0], o)), ("recover", m)
    okd = outs_d[0][1]
    assert 0 < int(okd.sum()) < m and okd[0] == 1
    for c in (c0, c1, c2, c3):
        assert PC.check_mul_golden(c, "secp256k1") > 50
        assert PC.check_decompress_golden(c, "secp256k1") > 40
        assert PC.check_recover_golden(c, "secp256k1") >= 30
        assert PC.check_exceptional_keys(c, "secp256k1") > 400
        assert PC.check_verify_golden(c, "secp256k1") > 15
        assert PC.check_offcurve_golden(c, "secp256k1") >= 29
    for c in (c0, c1, c2, c3, cd):
        c.close()


@pytest.mark.parametrize("curve", ["p256", "p224", "p192", "p384", "p521"])
def test_nist_small_batches_on_the_row_layer(monkeypatch, curve):
    """Batches of at most ELLGPU_COOP_GRID items on the NIST curves run every item's ladder and comb
    on a wave of its own -- up to 256 bits csrc/coop_mont.h (a Montgomery field of nine 29-bit limbs
    across a 16-lane DPP row), p384 / p521 csrc/coop_wide.h (14 / 19 28-bit digits across the whole
    wave: DPP wave_shr, ds_bpermute; round 6) -- coop_work.h CoopNist, joined by one-lane kernels.  Same bytes as the
    one-item-per-lane kernels (ELLGPU_COOP_GRID=0) for verify (valid, corrupted, off-curve keys, r
    or s out of range), P*k (k = 0, 1, n - 1, n, n + 1, 2^(8B) - 1 included), k1*G + k2*P, and the
    reference's fixtures through both."""
    B = elliptic_amd.FIELD_BYTES[curve]
    monkeypatch.setenv("ELLGPU_COOP_GRID", "0")
    c0 = elliptic_amd.Context(0)
    monkeypatch.setenv("ELLGPU_COOP_GRID", str(1 << 30))
    c1 = elliptic_amd.Context(0)
    monkeypatch.delenv("ELLGPU_COOP_GRID")
    cd = elliptic_amd.Context(0)
    cur = O.get_curve(curve)
    n = 1500
    raw = np.frombuffer(hashlib.shake_256(("row-layer:" + curve).encode()).digest(n * 4 * B), dtype=np.uint8).reshape(n, 4 * B)
    d, k, z, e = (np.ascontiguousarray(raw[:, i * B:(i + 1) * B]) for i in range(4))
    for i, kv in enumerate((0, 1, 2, cur.n - 1, cur.n, cur.n + 1, (1 << (8 * B)) - 1)):
        k[10 + i] = np.frombuffer(int(kv).to_bytes(B, "big"), np.uint8)
    pts, _ = c0.mul_fixed(curve, d)
    r, s_, rec, okk = c0.ecdsa_sign_det(curve, z, d)
    assert okk.all()
    zz = z.copy()
    zz[::7, 0] ^= 1                                      # corrupted digests
    rr = r.copy()
    rr[3] = 0                                            # r = 0
    bad = pts.copy()
    bad[5::11, 2 * B - 1] ^= 1                           # keys off the curve
    for m in (1, 2, 17, 64, 65, 300, 1365, 1366, 1500):
        outs = []
        for c, rowk in ((c0, False), (c1, True), (cd, m <= 1365)):
            c.set_timing(True)
            v = c.ecdsa_verify(curve, zz[:m], rr[:m], s_[:m], bad[:m], status=True)
            tm = c.get_timing()
            assert ("ecdsa_parts_c" in tm) == rowk and ("ecdsa_main" in tm) != rowk, (curve, m, rowk, sorted(tm))
            mv = c.mul_var(curve, k[:m], pts[:m])
            tm = c.get_timing()
            assert ("mul_parts_c" in tm) == rowk, (curve, m, rowk, sorted(tm))
            ma = c.mul_add2(curve, e[:m], None, k[:m], pts[:m])
            c.set_timing(True)                             # (a fresh recording)
            mf = c.mul_fixed(curve, k[:m])                 # the comb + the item's own inversion on a wave
            tm = c.get_timing()
            assert ("mul_fixed_c" in tm) == rowk and ("mul_fixed" in tm) != rowk, (curve, m, rowk, sorted(tm))
            c.set_timing(False)
            outs.append((v, mv, ma, mf))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (curve, m)
    ok, st = outs[0][0]
    assert ok[1::7][:50].sum() > 30 and not ok[::7].any() and (st[5::11] == 2).sum() > 100 and ok[3] == 0
    for c in (c0, c1):
        assert PC.check_verify_golden(c, curve) > 15 and PC.check_mul_golden(c, curve) > 50
        assert PC.check_offcurve_golden(c, curve) >= 29
        if curve != "p192":
            assert PC.check_exceptional_keys(c, curve) > 400
    for c in (c0, c1, cd):
        c.close()


def test_eddsa_verify_small_batches_on_the_row_layer(monkeypatch):
    """EDDSA#verify of at most ELLGPU_COOP_GRID items: the two sides of the equation on a wave each
    (csrc/coop_ed.h; eddsa_parts_c), compared by the one-lane eddsa_join -- against the one-kernel
    form (ELLGPU_COOP_GRID=0) on signatures of this library (messages of 0..200 bytes), corrupted
    messages / R / S / keys, S >= n, encodings that do not decode (the reference throws: err = 1),
    non-canonical y, and the reference's vectors through both."""
    monkeypatch.setenv("ELLGPU_COOP_GRID", "0")
    c0 = elliptic_amd.Context(0)
    monkeypatch.setenv("ELLGPU_COOP_GRID", str(1 << 30))
    c1 = elliptic_amd.Context(0)
    monkeypatch.delenv("ELLGPU_COOP_GRID")
    cd = elliptic_amd.Context(0)
    n = 1500
    raw = hashlib.shake_256(b"row-layer:eddsa").digest(n * 32 + n * 200)
    sec = np.frombuffer(raw[:n * 32], np.uint8).reshape(n, 32)
    msgs = [raw[n * 32 + 200 * i: n * 32 + 200 * i + (i * 7) % 201] for i in range(n)]
    sig, pub = c0.eddsa_sign(msgs, sec)
    sig, pub, msgs = sig.copy(), pub.copy(), list(msgs)
    L_ORDER = 2 ** 252 + 27742317777372353535851937790883648493
    for i in range(0, n, 9):                              # one defect per ninth item, by kind
        kind = (i // 9) % 8
        if kind == 0: msgs[i] = msgs[i] + b"x"
        elif kind == 1: sig[i, 3] ^= 0x10                 # R: another point, or no point at all
        elif kind == 2: sig[i, 40] ^= 1                   # S
        elif kind == 3: pub[i, 7] ^= 0x20                 # A
        elif kind == 4:                                   # S + n (still 32 bytes): S >= n -> false, nothing decoded
            S = int.from_bytes(sig[i, 32:].tobytes(), "little") + L_ORDER
            if S < 1 << 256: sig[i, 32:] = np.frombuffer(S.to_bytes(32, "little"), np.uint8)
        elif kind == 5: sig[i, :32] = np.frombuffer((2).to_bytes(32, "little"), np.uint8)       # y = 2: not on the curve
        elif kind == 6: pub[i] = np.frombuffer((2 ** 255 - 19 + 1).to_bytes(32, "little"), np.uint8)   # y = p + 1: non-canonical 1
        else: pub[i, 31] ^= 0x80                          # the other x
    for m in (1, 2, 33, 300, 1365, 1366, 1500):
        outs = []
        for c, rowk in ((c0, False), (c1, True), (cd, m <= 1365)):
            c.set_timing(True)
            outs.append(c.eddsa_verify(msgs[:m], sig[:m], pub[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("eddsa_parts_c" in tm) == rowk and ("eddsa_verify" in tm) != rowk, (m, rowk, sorted(tm))
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), m
    ok, err = outs[0]
    good = np.ones(n, bool)
    good[::9] = False
    assert ok[good].all() and not ok[::9].any() and not err[good].any() and err[::9].sum() > 10
    # edwards Point#mul / mulAdd of a handful of items: one item per wave (ed_mul_c)
    kk = np.frombuffer(hashlib.shake_256(b"row-layer:edmul").digest(n * 64), np.uint8).reshape(n, 64)
    k1, k2 = np.ascontiguousarray(kk[:, :32]), np.ascontiguousarray(kk[:, 32:])
    for i, kv in enumerate((0, 1, 2, L_ORDER - 1, L_ORDER, L_ORDER + 1, (1 << 256) - 1)):
        k2[20 + i] = np.frombuffer(int(kv).to_bytes(32, "big"), np.uint8)
    pts, _ = c0.mul_fixed("ed25519", sec)
    pts2, _ = c0.mul_fixed("ed25519", k1)
    for m in (1, 3, 64, 1365, 1366):
        outs = []
        for c, rowk in ((c0, False), (c1, True), (cd, m <= 1365)):
            c.set_timing(True)
            r = (c.mul_var("ed25519", k2[:m], pts[:m]), c.mul_add2("ed25519", k1[:m], pts2[:m], k2[:m], pts[:m]),
                 c.mul_add2("ed25519", k1[:m], None, k2[:m], pts[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("ed_mul_c" in tm) == rowk and ("ed_mul_var" in tm) != rowk, (m, rowk, sorted(tm))
            outs.append(r)
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), m
    # edwards Point#mul on G (the comb + the item's own inversion on a wave: ed_mul_fixed_c) and
    # EDDSA#sign, whose a*G and r*G are two such items: same bytes as ed_mul_fixed -> ed_normalize
    for m in (1, 3, 64, 682, 683, 1365, 1366):
        outs = []
        for c, rowk in ((c0, False), (c1, True), (cd, m <= 1365)):
            c.set_timing(True)
            mf = c.mul_fixed("ed25519", k2[:m])
            tm = c.get_timing()
            assert ("ed_mul_fixed_c" in tm) == rowk and ("ed_mul_fixed" in tm) != rowk, (m, rowk, sorted(tm))
            c.set_timing(True)
            sg = c.eddsa_sign(msgs[:m], sec[:m])
            tm = c.get_timing()
            c.set_timing(False)
            rows = rowk if c is not cd else 2 * m <= 1365
            assert ("ed_mul_fixed_c" in tm) == rows and ("ed_mul_fixed" in tm) != rows, (m, rowk, sorted(tm))
            outs.append((mf, sg))
        for o in outs[1:]:
            for a, b in zip(outs[0], o):
                assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), m
    assert outs[0][0][1][20] == 1 and outs[0][0][1][24] == 1 and outs[0][0][1][:20].sum() == 0   # k = 0 and k = n: the identity
    # curve25519 Point#mul (x only), one item per wave (x25519_c): unclamped scalars incl. 0, 1 and
    # 2^256 - 1, u = 0, 1, p - 1 and a point of small order
    xs = np.ascontiguousarray(kk[:, :32]).copy()
    xs[:, 0] &= 0x7F
    for i, xv in enumerate((0, 1, 2 ** 255 - 20, 9, 325606250916557431795983626356110631294008115727848805560023387167927233504)):
        xs[30 + i] = np.frombuffer(int(xv).to_bytes(32, "big"), np.uint8)
    for m in (1, 40, 1365, 1366):
        outs = []
        for c, rowk in ((c0, False), (c1, True), (cd, m <= 1365)):
            c.set_timing(True)
            outs.append(c.x25519(k2[:m], xs[:m]))
            tm = c.get_timing()
            c.set_timing(False)
            assert ("x25519_c" in tm) == rowk and ("x25519_ladder" in tm) != rowk, (m, rowk, sorted(tm))
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), m
    for c in (c0, c1):
        assert PC.check_eddsa_golden(c) > 200 and PC.check_mul_golden(c, "ed25519") > 50
        assert PC.check_x25519_golden(c) > 30
    for c in (c0, c1, cd):
        c.close()


@pytest.mark.parametrize("curve", ["secp256k1", "p256", "p224", "p192", "p384", "p521"])
def test_small_batch_sign_on_the_row_layer(monkeypatch, curve):
    """EC#sign for a handful of items: k*G (comb + the item's own inversion) and k^-1 mod n on a
    wave each in ONE launch of the row layer (sign_parts_c), then sign_finish without an inversion
    -- instead of sign_mul -> normalize -> sign_finish with its inversion.  Same (r, s, recovery
    parameter, ok) as the one-lane pipeline: deterministic nonces, supplied nonces incl. k = 0, 1,
    n - 1 (rejected), canonical form."""
    B, NB = elliptic_amd.FIELD_BYTES[curve], elliptic_amd.ORDER_BYTES[curve]
    monkeypatch.setenv("ELLGPU_COOP_GRID", "0")
    c0 = elliptic_amd.Context(0)
    monkeypatch.delenv("ELLGPU_COOP_GRID")
    c1 = elliptic_amd.Context(0)
    cur = O.get_curve(curve)
    n = 1400
    raw = np.frombuffer(hashlib.shake_256(("row-sign:" + curve).encode()).digest(n * 3 * NB), dtype=np.uint8).reshape(n, 3 * NB)
    d, z, kn = (np.ascontiguousarray(raw[:, i * NB:(i + 1) * NB]) for i in range(3))
    for i, kv in enumerate((0, 1, cur.n - 1, 2, cur.n - 2)):
        kn[20 + i] = np.frombuffer(int(kv).to_bytes(NB, "big"), np.uint8)
    for m in (1, 3, 64, 300, 1365, 1400):
        for canonical in (False, True):
            c1.set_timing(True)
            a = c1.ecdsa_sign_det(curve, z[:m], d[:m], canonical=canonical)
            tm = c1.get_timing()
            c1.set_timing(False)
            assert ("sign_parts_c" in tm) == (m <= 1365) and ("sign_mul" in tm) == (m > 1365), (curve, m, sorted(tm))
            b = c0.ecdsa_sign_det(curve, z[:m], d[:m], canonical=canonical)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (curve, m, canonical)
        a = c1.ecdsa_sign(curve, z[:m], d[:m], kn[:m])
        b = c0.ecdsa_sign(curve, z[:m], d[:m], kn[:m])
        assert all(np.array_equal(x, y) for x, y in zip(a, b)), (curve, m)
    if curve != "p521":
        assert not a[3][20:23].any() and a[3][23:25].all()       # k = 0, 1, n - 1 rejected; 2 and n - 2 signed
    else:
        # (a 66-byte nonce with a non-zero top byte goes through _truncateToN(K, true) first, ec/index.js:
        # 153-156: n - 1 and n - 2 arrive shifted right by 7 bits -- ordinary nonces, signed)
        assert not a[3][20:22].any() and a[3][22:25].all()
    assert PC.check_sign_golden(c1, curve) > 10 and PC.check_signdet_golden(c1, curve) > 10
    c0.close()
    c1.close()


def test_fixed_base_table_narrows_when_memory_is_short(ctx, monkeypatch):
    """The signed comb's window width travels with the table (ladder.h comb_bits_of): a context
    that cannot have the default 22-bit table (1.6 GB) builds a 16-bit one (36 MB; here forced
    with ELLGPU_COMB_MAX_BYTES) and the same kernels -- full-grid, small-grid, parted, row layer,
    sign, recover -- give the same bytes on it."""
    assert ctx.mul_fixed("secp256k1", np.ones((1, 32), np.uint8))[0].shape == (1, 64)
    assert ctx.comb_bits("secp256k1") == 22
    monkeypatch.setenv("ELLGPU_COMB_MAX_BYTES", str(100 << 20))
    c = elliptic_amd.Context(0)
    monkeypatch.delenv("ELLGPU_COMB_MAX_BYTES")
    n = 200000
    h, r, s, pub, expect = _make_sigs(ctx, n, "gpu-test-narrow-comb")
    assert np.array_equal(c.ecdsa_verify("secp256k1", h, r, s, pub), expect)          # full grid
    assert c.comb_bits("secp256k1") == 16
    for m in (1, 100, 5000, 70000):                                                  # row layer, parts, small grid
        assert np.array_equal(c.ecdsa_verify("secp256k1", h[:m], r[:m], s[:m], pub[:m]), expect[:m]), m
    ks = r.copy()
    ks[0] = 0
    ks[1] = 255
    for m in (n, 300):
        a, ai = c.mul_fixed("secp256k1", ks[:m])
        b, bi = ctx.mul_fixed("secp256k1", ks[:m])
        assert np.array_equal(a, b) and np.array_equal(ai, bi), m
        a, ai = c.mul_add2("secp256k1", ks[:m], None, s[:m], pub[:m])
        b, bi = ctx.mul_add2("secp256k1", ks[:m], None, s[:m], pub[:m])
        assert np.array_equal(a, b) and np.array_equal(ai, bi), m
    assert PC.check_signdet_golden(c, "secp256k1") > 10 and PC.check_recover_golden(c, "secp256k1") >= 30
    assert PC.check_verify_golden(c, "p256") > 15 and c.comb_bits("p256") == 16
    c.close()


def test_full_size_group_properties(ctx):
    """size-independent properties on 2^18 items: (a*G)*b == (b*G)*a, fixed ==
    variable base on G, k1*G + k2*G == (k1+k2)*G."""
    n = 1 << 18
    N = O.get_curve("secp256k1").n
    raw = np.frombuffer(hashlib.shake_256(b"gpu-prop").digest(n * 64), dtype=np.uint8).reshape(n, 64)
    a, b = np.ascontiguousarray(raw[:, :32]), np.ascontiguousarray(raw[:, 32:])
    A, _ = ctx.mul_fixed("secp256k1", a)
    Bp, _ = ctx.mul_fixed("secp256k1", b)
    AB, i1 = ctx.mul_var("secp256k1", b, A)
    BA, i2 = ctx.mul_var("secp256k1", a, Bp)
    assert np.array_equal(AB, BA) and np.array_equal(i1, i2)
    cur = O.get_curve("secp256k1")
    G = np.tile(ints_to_be([cur.g.x, cur.g.y], 32).reshape(1, 64), (n, 1))
    Av, _ = ctx.mul_var("secp256k1", a, G)
    assert np.array_equal(A, Av)
    m = 1 << 14
    S, _ = ctx.mul_add2("secp256k1", a[:m], None, b[:m], G[:m])
    sums = [(x + y) % N for x, y in zip(be_to_ints(a[:m]), be_to_ints(b[:m]))]
    S2, _ = ctx.mul_fixed("secp256k1", ints_to_be(sums, 32))
    assert np.array_equal(S, S2)


def test_chunk_boundary(ctx):
    """batches larger than the engine's 2^21-item launch chunk: the items either side of the
    cut must equal what separate calls give (fixed base, variable base and verify)"""
    n = (1 << 21) + 777
    raw = np.frombuffer(hashlib.shake_256(b"gpu-chunk").digest(n * 32), dtype=np.uint8).reshape(n, 32)
    xy, inf = ctx.mul_fixed("secp256k1", raw)
    lo, hi = (1 << 21) - 300, (1 << 21) + 300
    xy2, inf2 = ctx.mul_fixed("secp256k1", raw[lo:hi])
    assert np.array_equal(xy[lo:hi], xy2) and np.array_equal(inf[lo:hi], inf2)
    xy3, inf3 = ctx.mul_fixed("secp256k1", raw[-50:])
    assert np.array_equal(xy[-50:], xy3) and not inf.any()
    k = np.roll(raw, 1, axis=0)
    out, oinf = ctx.mul_var("secp256k1", k, xy)
    out2, oinf2 = ctx.mul_var("secp256k1", k[lo:hi], xy[lo:hi])
    assert np.array_equal(out[lo:hi], out2) and np.array_equal(oinf[lo:hi], oinf2)
    cur = O.get_curve("secp256k1")
    for i in (0, (1 << 21) - 1, 1 << 21, n - 1):
        w = cur.g.mul(int.from_bytes(raw[i].tobytes(), "big")).mul(int.from_bytes(k[i].tobytes(), "big"))
        assert (int.from_bytes(out[i, :32].tobytes(), "big"), int.from_bytes(out[i, 32:].tobytes(), "big")) == (w.x, w.y)


@pytest.mark.parametrize("idx", range(5))
def test_user_defined_short_curves_gpu(ctx, idx):
    """`new curve.short({p, a, b})` with parameters that are no preset (the
    generic-a `_dbl` / `dblp`): Point#mul, mulAdd / jmulAdd, Point#add against the reference's
    results (tests/golden/custom_short.json)"""
    spec = PC.custom_curves()[idx]
    assert PC.check_custom_short_golden(ctx, spec) > 80


@pytest.mark.parametrize("idx", range(4))
def test_user_defined_edwards_curves_gpu(ctx, idx):
    """`new curve.edwards({p, a, c: 1, d})` with parameters that are not ed25519's (SURVEY.md 8 row
    a16: _projDbl / _projAdd): the reference's results on four such curves"""
    assert PC.check_custom_edwards_golden(ctx, PC.custom_edwards_curves()[idx]) > 80


def test_dev_calls_on_alternating_streams(ctx):
    """*_dev calls work in the context's scratch arenas: issued from several torch streams without
    any host synchronisation in between (they used to race on the window tables), every
    call's result must equal the one-stream result.  Two streams get an arena each (two passes in
    flight); a third stream takes the arena used longest ago, behind the event its previous user
    recorded (HipBackend::use_stream_dev / end_call); a host-buffer call in between waits for all
    of them."""
    import torch
    dev = torch.device("cuda", 0)
    n = 1 << 16
    rng = np.random.default_rng(7)
    ks = [rng.integers(0, 256, (n, 32), dtype=np.uint8) for _ in range(6)]
    g = np.frombuffer(O.get_curve("secp256k1").g.x.to_bytes(32, "big") + O.get_curve("secp256k1").g.y.to_bytes(32, "big"), np.uint8)
    pts, _ = ctx.mul_fixed("secp256k1", ks[0])
    want = [ctx.mul_var("secp256k1", k, pts) for k in ks]
    dk = [torch.from_numpy(k).to(dev) for k in ks]
    dp = torch.from_numpy(pts).to(dev)
    outs = [(torch.zeros(n, 64, dtype=torch.uint8, device=dev), torch.zeros(n, dtype=torch.uint8, device=dev)) for _ in ks]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize()                        # the zero_() calls below run on the default stream
        for i, k in enumerate(dk):
            with torch.cuda.stream(streams[i & 1]):
                ctx.mul_var_dev("secp256k1", k, dp, outs[i][0], outs[i][1])
        torch.cuda.synchronize()
        for i in range(len(ks)):
            assert np.array_equal(outs[i][0].cpu().numpy(), want[i][0]) and np.array_equal(outs[i][1].cpu().numpy(), want[i][1]), (rep, i)
            outs[i][0].zero_()
    # three streams round-robin, batch sizes of every tuning (parted, small-grid, full-grid), a
    # host-buffer call in the middle of the queue, and verifies between the multiplications
    streams.append(torch.cuda.Stream(device=dev))
    h, r, s, pub, expect = _make_sigs(ctx, n, "gpu-test-streams")
    dv = [torch.from_numpy(x).to(dev) for x in (h, r, s, pub)]
    sizes = [n, 100, 20000, n, 3000, 50000]
    oks = [torch.zeros(m, dtype=torch.uint8, device=dev) for m in sizes]
    torch.cuda.synchronize()
    for rep in range(3):
        torch.cuda.synchronize()                        # as above: zero_() is queued on the default stream
        for i, k in enumerate(dk):
            m = sizes[i]
            with torch.cuda.stream(streams[i % 3]):
                ctx.mul_var_dev("secp256k1", k[:m], dp[:m], outs[i][0][:m], outs[i][1][:m])
                ctx.ecdsa_verify_dev("secp256k1", *(x[:m] for x in dv), oks[i])
            if i == 2:
                hx, hi = ctx.mul_var("secp256k1", ks[1][:777], pts[:777])          # host buffers: waits for the lanes
                assert np.array_equal(hx, want[1][0][:777]) and np.array_equal(hi, want[1][1][:777])
        torch.cuda.synchronize()
        for i, m in enumerate(sizes):
            assert np.array_equal(outs[i][0][:m].cpu().numpy(), want[i][0][:m]) and np.array_equal(outs[i][1][:m].cpu().numpy(), want[i][1][:m]), (rep, i)
            assert np.array_equal(oks[i].cpu().numpy(), expect[:m]), (rep, i)
            outs[i][0].zero_()
            oks[i].zero_()


def test_sharded_helpers_on_device(ctx):
    """elliptic_amd.sharding on a CUDA device (world size 1: the slicing, the stream-ordered
    launch without host synchronisation, device-resident and host inputs)"""
    import torch
    from elliptic_amd.sharding import ShardedMul, ShardedVerifier
    from golden_util import I, verify_cases
    dev = torch.device("cuda", 0)
    cs = [c for c in verify_cases("secp256k1") if len(c["z"]) == 64 and len(c["r"]) <= 64 and len(c["s"]) <= 64]
    z = np.frombuffer(b"".join(bytes.fromhex(c["z"]) for c in cs), np.uint8).reshape(-1, 32)
    r = ints_to_be([I(c["r"]) % (1 << 256) for c in cs], 32)
    s = ints_to_be([I(c["s"]) % (1 << 256) for c in cs], 32)
    q = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], 32), ints_to_be([I(c["qy"]) for c in cs], 32)], axis=1)
    want = np.array([1 if c["ok"] else 0 for c in cs], np.uint8)
    sv = ShardedVerifier(ctx, "secp256k1", dist=None, device=dev)
    assert np.array_equal(sv.verify(z, r, s, q).cpu().numpy(), want)
    tz, tr, ts, tq = (torch.from_numpy(a).to(dev) for a in (z, r, s, q))
    assert np.array_equal(sv.verify(tz, tr, ts, tq).cpu().numpy(), want)
    k = ints_to_be([5, 7, 0, 11], 32)
    xy, inf = ShardedMul(ctx, "secp256k1", dist=None, device=dev).mul(k)
    wxy, winf = ctx.mul_fixed("secp256k1", k)
    assert np.array_equal(xy.cpu().numpy(), wxy) and np.array_equal(inf.cpu().numpy(), winf)


@pytest.mark.gpu
def test_deferred_small_calls_gpu(ctx):
    """ellgpu_ctx_defer / ellgpu_ctx_collect on the device: the call returns with its work in flight,
    collect() -- or any other entry point -- completes it; same bytes as the plain call"""
    PC.check_deferred_calls(ctx)


@pytest.mark.gpu
def test_x25519_derive_gpu(ctx):
    """ellgpu_x25519_derive on the device: the validity test on a wave of its own beside the ladder's
    (x25519_c, one launch) for a handful of items, one-lane kernels above that; statuses against
    Python's Euler criterion, secrets against ellgpu_x25519_ladder"""
    assert PC.check_x25519_derive(ctx) >= 100
