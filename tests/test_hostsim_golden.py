"""CPU-only checks of the DEVICE CODE's logic: tests/hostsim compiles the same
headers the HIP kernels are built from with g++ and runs every thread in a
loop.  These tests drive that build through the same C ABI as libellgpu.so and
compare with the reference's golden vectors.  (The GPU parity tests proper are
in test_gpu_parity.py, -m gpu.)"""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from hostsim.build import build as build_hostsim  # noqa: E402

import elliptic_amd  # noqa: E402
from elliptic_amd import _lib  # noqa: E402
import parity_checks as PC  # noqa: E402
from oracle import ec_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def hs():
    path = build_hostsim()
    lib = _lib.load(path, optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing", "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
    return lib


@pytest.fixture(scope="module")
def ctx(hs):
    c = elliptic_amd.Context(0, lib_path=hs)
    yield c
    c.close()


FIELDS = {
    0: int(O.get_curve("secp256k1", False).p), 1: 2 ** 255 - 19, 2: int(O.get_curve("secp256k1", False).p),
    3: int(O.get_curve("secp256k1", False).p),        # the lanes-per-item field (csrc/coop.h), host simulation of the row
    10: O.get_curve("secp256k1", False).p, 11: O.get_curve("p192", False).p,
    12: O.get_curve("p224", False).p, 13: O.get_curve("p256", False).p,
    14: O.get_curve("p384", False).p, 15: O.get_curve("p521", False).p,
    20: O.get_curve("secp256k1", False).n, 21: O.get_curve("p192", False).n,
    22: O.get_curve("p224", False).n, 23: O.get_curve("p256", False).n,
    24: O.get_curve("p384", False).n, 25: O.get_curve("p521", False).n,
    26: O.get_curve("ed25519", False).n,
    # the row layer's Montgomery fields (csrc/coop_mont.h: nine 29-bit limbs across a 16-lane row, R = 2^261)
    31: O.get_curve("p192", False).p, 32: O.get_curve("p224", False).p, 33: O.get_curve("p256", False).p,
    # the WIDE fields (csrc/coop_wide.h: 14 / 19 28-bit digits over the lanes of a wave, R = 2^392 / 2^532)
    34: O.get_curve("p384", False).p, 35: O.get_curve("p521", False).p,
}


def _limbs(x, L):
    return (ctypes.c_uint32 * L)(*[(x >> (32 * i)) & 0xFFFFFFFF for i in range(L)])


def _val(arr):
    return sum(int(v) << (32 * i) for i, v in enumerate(arr))


@pytest.mark.parametrize("field", sorted(FIELDS))
def test_field_ops(hs, field):
    p = FIELDS[field]
    L = hs.hs_field_limbs(field)
    rnd = random.Random(1000 + field)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 2 ** (32 * L) - 1 if field in (0, 1) else p - 3,
            2 ** 32 - 1, 2 ** 64, p >> 1]
    vals = [(a % (2 ** (32 * L))) for a in edge] + [rnd.randrange(p) for _ in range(60)]
    # products that land in [p, 2^(32L)) before the final (rarely taken) subtraction
    import math
    for a, b in [(2, (p + 1) // 2), (3, (p + 2) // 3), (math.isqrt(p) + 1, math.isqrt(p) + 1),
                 (math.isqrt(p) + 1, math.isqrt(p) + 2), (p - 1, p - 1), (p - 1, 2)]:
        for op, fn in ((2, lambda x, y: x * y % p), (3, lambda x, y: x * x % p)):
            r = (ctypes.c_uint32 * L)()
            assert hs.hs_field_op(field, op, _limbs(a, L), _limbs(b, L), r) == 0
            assert _val(r) == fn(a % p, b % p), (field, op, hex(a), hex(b))
    for a in vals:
        for b in (vals[rnd.randrange(len(vals))], rnd.randrange(p), p - 1):
            for op, fn in ((0, lambda x, y: (x + y) % p), (1, lambda x, y: (x - y) % p),
                           (2, lambda x, y: x * y % p), (3, lambda x, y: x * x % p),
                           (5, lambda x, y: (-x) % p)):
                r = (ctypes.c_uint32 * L)()
                assert hs.hs_field_op(field, op, _limbs(a, L), _limbs(b, L), r) == 0
                assert _val(r) == fn(a % p, b % p), (field, op, hex(a), hex(b))
    for a in vals[:20]:
        if a % p == 0 or field == 3:                  # (the row layer has no inversion: the join kernels invert)
            continue
        r = (ctypes.c_uint32 * L)()
        hs.hs_field_op(field, 4, _limbs(a, L), _limbs(0, L), r)
        assert _val(r) == pow(a % p, -1, p)


def _inv_values(p, rnd, count):
    vals = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 30, (1 << 30) - 1, (1 << 60) + 1,
            1 << (p.bit_length() - 1), (1 << (p.bit_length() - 1)) - 1, p >> 1, 0x3FFFFFFF << 30]
    vals += [(1 << k) % p for k in range(1, p.bit_length(), 29)]
    vals += [(rnd.randrange(p) >> k) << k for k in (1, 17, 30, 31, 61, 90)]       # long runs of even steps
    vals += [rnd.randrange(p) for _ in range(count)]
    return [v % p for v in vals]


@pytest.mark.parametrize("field", sorted(FIELDS))
def test_field_inversion(hs, field):
    """division-step inversion (csrc/safegcd.h) in every field: edge values, powers of two,
    operands with long zero runs, random; inv(0) = 0 as bn.js `invm` callers expect"""
    if field == 3:
        pytest.skip("the lanes-per-item field has no inversion")
    p = FIELDS[field]
    L = hs.hs_field_limbs(field)
    rnd = random.Random(31337 + field)
    for a in _inv_values(p, rnd, 150):
        r = (ctypes.c_uint32 * L)()
        assert hs.hs_field_op(field, 4, _limbs(a, L), _limbs(0, L), r) == 0
        assert _val(r) == (pow(a, -1, p) if a else 0), (field, hex(a))


def test_rare_branches_k256_25519(hs):
    """Directed vectors for the carry-ripple / final-subtraction branches that random inputs
    reach with probability ~2^-31 (FpK256 and Fp25519 add, sub, mul, sqr)."""
    import field_vectors
    vecs = field_vectors.rare_vectors() + field_vectors.shift_vectors() + field_vectors.fold_vectors() + field_vectors.fold25519_vectors()
    assert len(vecs) > 1500
    for field, op, a, b, want in vecs:
        r = (ctypes.c_uint32 * 8)()
        assert hs.hs_field_op(field, op, _limbs(a, 8), _limbs(b, 8), r) == 0
        assert _val(r) == want, (field, op, hex(a), hex(b))


def test_rare_branches_solinas(hs):
    """operands that drive the p256 / p384 lazy-accumulator fold into its rare branch"""
    import field_vectors
    vecs = field_vectors.solinas_vectors() + field_vectors.solinas_addsub_vectors() + field_vectors.p521_addsub_vectors()
    assert len(vecs) > 300
    for field, op, a, b, want in vecs:
        L = {11: 6, 12: 7, 13: 8, 14: 12, 15: 17}[field]
        r = (ctypes.c_uint32 * L)()
        assert hs.hs_field_op(field, op, _limbs(a, L), _limbs(b, L), r) == 0
        assert _val(r) == want, (field, op, hex(a), hex(b))


@pytest.mark.parametrize("curve", ["secp256k1", "p256"])
def test_signed_comb_window_boundaries(ctx, curve):
    """the signed comb's recoding (8-bit windows in this build, 22-bit on the device): scalars whose
    windows sit on the boundaries -- largest positive digit, first negative one, carries through
    several windows and into the carry window -- G*k against the Python oracle"""
    from elliptic_amd import ints_to_be
    ks = PC.comb_boundary_scalars(256, 8, 160)
    xy, inf = ctx.mul_fixed(curve, ints_to_be(ks, 32))
    cur = O.get_curve(curve)
    for i, k in enumerate(ks):
        w = cur.g.mul(k)
        got = None if inf[i] else (int.from_bytes(xy[i, :32].tobytes(), "big"), int.from_bytes(xy[i, 32:].tobytes(), "big"))
        assert got == (None if w.inf else (w.x, w.y)), hex(k)


def test_solinas_chain_fold_raw_words(hs):
    """FpSolinas::reduce_wide_chain (p192 / p224 / p384) on raw 2L-word values: the rarely taken
    ripple of every chain, the top-word test, near misses and random values (field-op probe 13)"""
    import field_vectors
    vecs = field_vectors.solinas_chain_vectors()
    assert len(vecs) > 300
    for field, op, lo, hi, want in vecs:
        L = {11: 6, 12: 7, 14: 12}[field]
        r = (ctypes.c_uint32 * L)()
        assert hs.hs_field_op(field, op, _limbs(lo, L), _limbs(hi, L), r) == 0
        assert _val(r) == want, (field, hex(lo), hex(hi))


def test_p521_from_plain_overrange(hs):
    """The Mersenne fold must canonicalise any 17-limb input (decompress hands raw 66-byte
    values to from_plain): p -> 0, 2^521 -> 1, all-ones limbs, ..."""
    p = FIELDS[15]
    L = 17
    rnd = random.Random(77)
    vals = [p, p + 1, 2 * p, 2 * p + 1, 2 ** 521, 2 ** 522 - 1, 2 ** 528 - 1, 2 ** 544 - 1,
            2 ** 544 - 2 ** 521, (2 ** 23 - 1) * p, (2 ** 23 - 1) * p + p - 1]
    vals += [rnd.getrandbits(544) for _ in range(200)]
    for a in vals:
        r = (ctypes.c_uint32 * L)()
        assert hs.hs_field_op(15, 9, _limbs(a, L), _limbs(0, L), r) == 0
        assert _val(r) == a % p, hex(a)
        assert hs.hs_field_op(15, 2, _limbs(a, L), _limbs(a, L), r) == 0
        assert _val(r) == a * a % p, hex(a)


def _py_hmac_drbg(hname, entropy, nonce):
    """hmac-drbg 1.0.1 (lib/hmac-drbg.js:37-113) in Python: a generator of generate(len) calls"""
    import hashlib
    import hmac as pyhmac
    H = getattr(hashlib, hname)
    outlen = H().digest_size
    state = {"K": b"\x00" * outlen, "V": b"\x01" * outlen}

    def upd(seed):
        state["K"] = pyhmac.new(state["K"], state["V"] + b"\x00" + seed, H).digest()
        state["V"] = pyhmac.new(state["K"], state["V"], H).digest()
        if seed:
            state["K"] = pyhmac.new(state["K"], state["V"] + b"\x01" + seed, H).digest()
            state["V"] = pyhmac.new(state["K"], state["V"], H).digest()

    upd(entropy + nonce)

    def generate(n):
        out = b""
        while len(out) < n:
            state["V"] = pyhmac.new(state["K"], state["V"], H).digest()
            out += state["V"]
        upd(b"")
        return out[:n]
    return generate


@pytest.mark.parametrize("kind,hname,nbytes", [(0, "sha256", 24), (0, "sha256", 28), (0, "sha256", 32), (1, "sha384", 48),
                                               (2, "sha512", 66), (3, "sha256", 32), (3, "sha256", 21)])
def test_drbg_draws_in_a_row(hs, kind, hname, nbytes):
    """EC#sign's nonce generators (ec/index.js:141-158 -> hmac-drbg.js), FIVE draws in a row: a
    signature takes the first candidate in all but ~2^-128 of the cases, so the reseed in front of
    a further draw (the _update() that ends generate, deferred to where it is needed) and the
    constant key states of the all-zero K are checked here, against Python's hmac"""
    import random
    hs.hs_drbg_draws.restype = ctypes.c_int
    rnd = random.Random(1000 * kind + nbytes)
    for _ in range(4):
        seed = bytes(rnd.randrange(256) for _ in range(2 * nbytes))
        gen = _py_hmac_drbg(hname, seed[:nbytes], seed[nbytes:])
        want = b"".join(gen(nbytes) for _ in range(5))
        out = (ctypes.c_ubyte * (5 * nbytes))()
        assert hs.hs_drbg_draws(kind, (ctypes.c_ubyte * len(seed))(*seed), nbytes, 5, out) == 0
        assert bytes(out) == want


def test_glv_split(hs):
    cur = O.get_curve("secp256k1", False)
    lam, n = cur.endo["lambda"], cur.n
    rnd = random.Random(5)
    ks = [0, 1, 2, n - 1, n, n + 1, 2 ** 256 - 1, lam, lam + 1, n - lam, 2 ** 128, 2 ** 255]
    ks += [rnd.getrandbits(256) for _ in range(3000)]
    worst = 0
    for k in ks:
        k1 = (ctypes.c_uint32 * 5)()
        k2 = (ctypes.c_uint32 * 5)()
        n1, n2 = ctypes.c_int(), ctypes.c_int()
        hs.hs_glv_split(_limbs(k, 8), k1, ctypes.byref(n1), k2, ctypes.byref(n2))
        a, b = _val(k1), _val(k2)
        if n1.value:
            a = -a
        if n2.value:
            b = -b
        assert (a + b * lam - k) % n == 0
        worst = max(worst, abs(a).bit_length(), abs(b).bit_length())
    # the 33-nibble recoding without a carry window needs |k| < 2^131
    assert worst <= 129, worst


def test_glv_split_odd(hs):
    """glv_split<true>: the halves the secp256k1 ladders recode -- both odd (made so with the
    lattice vectors, not with a +1 that a later addition has to undo), still k1 + k2 lambda == k,
    and below 2^130 (the 33-window odd recoding takes < 2^132)"""
    cur = O.get_curve("secp256k1", False)
    lam, n = cur.endo["lambda"], cur.n
    rnd = random.Random(6)
    ks = [0, 1, 2, 3, n - 1, n, n + 1, 2 ** 256 - 1, 2 ** 256 - 2, lam, lam + 1, lam - 1, n - lam, 2 ** 128, 2 ** 255,
          2 ** 129, 2 ** 129 + 1]
    ks += [rnd.getrandbits(256) for _ in range(6000)]
    ks += [rnd.getrandbits(rnd.randrange(1, 257)) for _ in range(2000)]
    worst = 0
    seen = set()
    for k in ks:
        k1 = (ctypes.c_uint32 * 5)()
        k2 = (ctypes.c_uint32 * 5)()
        n1, n2 = ctypes.c_int(), ctypes.c_int()
        hs.hs_glv_split_odd(_limbs(k, 8), k1, ctypes.byref(n1), k2, ctypes.byref(n2))
        a, b = _val(k1), _val(k2)
        assert a & 1 and b & 1, hex(k)
        if n1.value:
            a = -a
        if n2.value:
            b = -b
        assert (a + b * lam - k) % n == 0, hex(k)
        worst = max(worst, abs(a).bit_length(), abs(b).bit_length())
        seen.add((n1.value, n2.value))
    assert worst <= 130, worst
    assert len(seen) == 4                      # every sign combination occurs


def test_recode_w4(hs):
    rnd = random.Random(9)
    for k in [0, 1, 7, 8, 9, 2 ** 256 - 1, 2 ** 255, 0x8888 << 240] + [rnd.getrandbits(256) for _ in range(300)]:
        d = (ctypes.c_byte * 65)()
        hs.hs_recode64(_limbs(k, 8), d)
        ds = list(d)
        assert all(-8 <= x <= 7 for x in ds[:64]) and ds[64] in (0, 1)
        assert sum(x * 16 ** i for i, x in enumerate(ds)) == k


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_mul_golden(ctx, curve):
    assert PC.check_mul_golden(ctx, curve) > 50


@pytest.mark.parametrize("curve", O.SHORT_CURVES + ["ed25519"])
def test_offcurve_operands_are_reported_not_guessed(ctx, curve):
    assert PC.check_offcurve_golden(ctx, curve) >= 29


@pytest.mark.parametrize("curve", ["secp256k1", "p256", "p224"])
def test_exceptional_keys_and_scalars(ctx, curve):
    assert PC.check_exceptional_keys(ctx, curve) > 400


def test_x25519_golden(ctx):
    assert PC.check_x25519_golden(ctx) > 30


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_verify_golden(ctx, curve):
    assert PC.check_verify_golden(ctx, curve) > 15


def _fresh_ctx(hs, monkeypatch, **env):
    """a context created under the given tuning overrides (they are read once, at creation)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    c = elliptic_amd.Context(0, lib_path=hs)
    for k in env:
        monkeypatch.delenv(k)
    return c


def test_verify_golden_secp256k1_both_tunings(hs, monkeypatch):
    """ecdsa_main exists in two tunings for secp256k1 (engine.h: FnEcdsaMain<.., WIDE>): the
    full-grid one (lean registers) and the small-grid one (entries requested one step ahead).
    Small batches take the second by default; ELLGPU_SMALL_GRID=0 (read when the context is
    created) forces the first."""
    c = _fresh_ctx(hs, monkeypatch, ELLGPU_SMALL_GRID="0")
    assert PC.check_verify_golden(c, "secp256k1") > 15
    assert PC.check_mul_golden(c, "secp256k1") > 50          # mul_var has the two tunings too
    assert PC.check_offcurve_golden(c, "secp256k1") >= 29
    c.close()
    c = _fresh_ctx(hs, monkeypatch, ELLGPU_SMALL_GRID=str(1 << 30))
    assert PC.check_verify_golden(c, "secp256k1") > 15
    assert PC.check_mul_golden(c, "secp256k1") > 50
    c.close()
    # the small-grid verify itself has two forms: prep || table -> ladder (default, the two
    # kernels of Work::ecdsa_table / ecdsa_ladder) and prep -> ecdsa_main (ELLGPU_SPLIT_VERIFY=0)
    c = _fresh_ctx(hs, monkeypatch, ELLGPU_SPLIT_VERIFY="0")
    assert PC.check_verify_golden(c, "secp256k1") > 15
    assert PC.check_offcurve_golden(c, "secp256k1") >= 29
    c.close()
    # ... and batches far below one wave per SIMD take the PARTED form by default (three lanes per
    # item: Work::ecdsa_half x 2, ecdsa_fixed, ecdsa_join); ELLGPU_PARTED_GRID=0 keeps them on the
    # ladder.  The parts themselves have two forms: one item per lane (work.h) and, for batches of
    # at most ELLGPU_COOP_GRID items, one item per WAVE with its field elements spread over a
    # 16-lane row (coop.h / coop_work.h; here the host simulation of the row) -- same join kernels.
    # Round 6: between the two, batches of at most ELLGPU_ROW_GRID items run one item per ROW of the
    # wave, four items per wave (coop.h FpK256R / coop_work.h CoopK256R: ecdsa_prep_table_r, ecdsa_parts_r,
    # mul_parts_r; the host passes walk the four rows one after the other).
    big = str(1 << 30)
    for grid, coop, row, parts in (("0", "0", "0", None), (big, "0", "0", b"ecdsa_parts"), (big, big, "0", b"ecdsa_parts_c"),
                                   (big, "0", big, b"ecdsa_parts_r")):
        c = _fresh_ctx(hs, monkeypatch, ELLGPU_PARTED_GRID=grid, ELLGPU_COOP_GRID=coop, ELLGPU_ROW_GRID=row, ELLGPU_ROW_FROM="0")
        mparts = {None: None, b"ecdsa_parts": b"mul_parts", b"ecdsa_parts_c": b"mul_parts_c", b"ecdsa_parts_r": b"mul_parts_r"}[parts]
        hs.hs_launches_reset()
        assert PC.check_verify_golden(c, "secp256k1") > 15
        assert PC.check_offcurve_golden(c, "secp256k1") >= 29
        assert PC.check_exceptional_keys(c, "secp256k1") > 400
        for name in (b"ecdsa_parts", b"ecdsa_parts_c", b"ecdsa_parts_r"):
            assert (hs.hs_launches(name) > 0) == (name == parts), name
        assert (hs.hs_launches(b"ecdsa_join") > 0) == (parts is not None)
        assert (hs.hs_launches(b"ecdsa_main") > 0) == (parts is None)
        # (the row layer also builds Q's window table, beside the prep: one launch)
        assert (hs.hs_launches(b"ecdsa_prep_table_c") > 0) == (parts == b"ecdsa_parts_c")
        assert (hs.hs_launches(b"ecdsa_prep_table_r") > 0) == (parts == b"ecdsa_parts_r")
        assert (hs.hs_launches(b"ecdsa_prep_table") > 0) == (parts not in (b"ecdsa_parts_c", b"ecdsa_parts_r"))
        # Point#mul likewise (Work::mul_half x 2, mul_join); k1 G + k2 P stays on its one ladder
        hs.hs_launches_reset()
        assert PC.check_mul_golden(c, "secp256k1") > 50
        for name in (b"mul_parts", b"mul_parts_c", b"mul_parts_r"):
            assert (hs.hs_launches(name) > 0) == (name == mparts), name
        assert (hs.hs_launches(b"mul_join") > 0) == (parts is not None)
        # fixed base: the comb and the item's own inversion on a wave (mul_fixed_c), or mul_fixed -> normalize
        assert (hs.hs_launches(b"mul_fixed_c") > 0) == (parts == b"ecdsa_parts_c")
        assert (hs.hs_launches(b"mul_fixed") > 0) == (parts != b"ecdsa_parts_c")
        # k1 G + k2 P (Point#mulAdd with G, EC#recoverPubKey): the halves of k2 and the comb of k1
        hs.hs_launches_reset()
        assert PC.check_exceptional_keys(c, "secp256k1") > 400
        assert PC.check_recover_golden(c, "secp256k1") >= 30
        assert (hs.hs_launches(b"mul_add_g") > 0) == (parts is None)
        for name in (b"mul_parts", b"mul_parts_c", b"mul_parts_r"):
            assert (hs.hs_launches(name) > 0) == (name == mparts), name
        # ... whose front -- R's square root and r^-1 -- is one launch of the row layer, side by side
        assert (hs.hs_launches(b"recover_parts_c") > 0) == (parts == b"ecdsa_parts_c")
        assert (hs.hs_launches(b"recover_prep") > 0) == (parts != b"ecdsa_parts_c")
        # ShortCurve#pointFromX of a handful of abscissas: the square root on a wave per item
        hs.hs_launches_reset()
        assert PC.check_decompress_golden(c, "secp256k1") > 40
        assert (hs.hs_launches(b"decompress_c") > 0) == (parts == b"ecdsa_parts_c")
        assert (hs.hs_launches(b"decompress") > 0) == (parts != b"ecdsa_parts_c")
        c.close()
    # the default context routes the lone call to the row layer
    c = elliptic_amd.Context(0, lib_path=hs)
    hs.hs_launches_reset()
    assert PC.check_verify_golden(c, "secp256k1") > 15
    assert hs.hs_launches(b"ecdsa_parts_c") > 0 and hs.hs_launches(b"ecdsa_parts") == 0
    c.close()


@pytest.mark.parametrize("curve", ["p192", "p224", "p256", "p384", "p521"])
def test_nist_curves_one_lane_and_row_layer(hs, monkeypatch, curve):
    """Small batches on the NIST curves run their ladder and comb on the lanes-per-item layer, one item
    per wave -- up to 256 bits csrc/coop_mont.h (a Montgomery field of nine 29-bit limbs across a
    16-lane row), p384 / p521 csrc/coop_wide.h (14 / 19 28-bit digits across the wave; round 6) --
    work: coop_work.h CoopNist, joined by the one-lane ecdsa_join2 / mul_join;
    ELLGPU_COOP_GRID=0 keeps them on the one-item-per-lane kernels.  Same results from both."""
    wide = curve in ("p384", "p521")
    # (the wide curves: the row-layer pass only, without the 400 exceptional keys -- the host simulation of
    # a 64-lane wave is slow, their one-lane kernels are covered by the other tests of this file, and the
    # GPU suite runs both forms on everything: test_nist_small_batches_on_the_row_layer)
    for coop, rowk in ((str(1 << 30), True),) if wide else (("0", False), (str(1 << 30), True)):
        c = _fresh_ctx(hs, monkeypatch, ELLGPU_COOP_GRID=coop)
        hs.hs_launches_reset()
        assert PC.check_verify_golden(c, curve) > 15
        assert PC.check_offcurve_golden(c, curve) >= 29
        if curve != "p192" and not wide:
            assert PC.check_exceptional_keys(c, curve) > 400
        assert (hs.hs_launches(b"ecdsa_parts_c") > 0) == rowk and (hs.hs_launches(b"ecdsa_main") > 0) != rowk
        # (the key's window table is built beside the scalar-field prep, one launch in front)
        assert (hs.hs_launches(b"ecdsa_prep_table_c") > 0) == rowk and (hs.hs_launches(b"ecdsa_prep") > 0) != rowk
        hs.hs_launches_reset()
        assert PC.check_mul_golden(c, curve) > 50
        assert (hs.hs_launches(b"mul_parts_c") > 0) == rowk
        assert (hs.hs_launches(b"mul_fixed_c") > 0) == rowk and (hs.hs_launches(b"mul_fixed") > 0) != rowk
        if curve != "p224":
            assert PC.check_recover_golden(c, curve) >= 30
        assert PC.check_sign_golden(c, curve) > 10
        c.close()


def test_eddsa_verify_one_lane_and_row_layer(hs, monkeypatch):
    """EDDSA#verify of at most ELLGPU_COOP_GRID items runs the two sides of its equation on a wave
    each (csrc/coop_ed.h: 2^255 - 19 as nine signed limbs across a DPP row, the four products of a
    step of the extended-coordinate formulas side by side in the wave's four rows) and compares them
    in the one-lane eddsa_join; ELLGPU_COOP_GRID=0 keeps the one-kernel form.  Same verdicts and
    'reference throws' flags from both, on the reference's vectors and on the edge encodings."""
    for coop, rowk in (("0", False), (str(1 << 30), True)):
        c = _fresh_ctx(hs, monkeypatch, ELLGPU_COOP_GRID=coop)
        hs.hs_launches_reset()
        assert PC.check_eddsa_golden(c) > 200
        assert (hs.hs_launches(b"eddsa_parts_c") > 0) == rowk and (hs.hs_launches(b"eddsa_verify") > 0) != rowk
        assert (hs.hs_launches(b"eddsa_join") > 0) == rowk
        # edwards Point#mul / mulAdd (all three forms) of a handful of items: one item per wave
        hs.hs_launches_reset()
        assert PC.check_mul_golden(c, "ed25519") > 50
        assert PC.check_offcurve_golden(c, "ed25519") >= 29
        assert (hs.hs_launches(b"ed_mul_c") > 0) == rowk and (hs.hs_launches(b"ed_mul_var") > 0) != rowk
        # ... on G: the comb and the item's own inversion on a wave (EDDSA#sign's a*G and r*G too)
        assert (hs.hs_launches(b"ed_mul_fixed_c") > 0) == rowk and (hs.hs_launches(b"ed_mul_fixed") > 0) != rowk
        hs.hs_launches_reset()
        assert PC.check_eddsa_sign_golden(c) > 100
        assert (hs.hs_launches(b"ed_mul_fixed_c") > 0) == rowk and (hs.hs_launches(b"ed_mul_fixed") > 0) != rowk
        # curve25519's x-only ladder, one item per wave (coop_ed.h CoopX25519)
        hs.hs_launches_reset()
        assert PC.check_x25519_golden(c) > 30
        assert (hs.hs_launches(b"x25519_c") > 0) == rowk and (hs.hs_launches(b"x25519_ladder") > 0) != rowk
        c.close()


@pytest.mark.parametrize("curve", ["secp256k1", "p192", "p256", "p384", "p521", "ed25519"])
def test_decompress_golden(ctx, curve):
    assert PC.check_decompress_golden(ctx, curve) > 40


def test_edwards_point_from_x_golden(ctx):
    assert PC.check_ed_from_x_golden(ctx) > 100


def test_decompress_p224_tonelli_shanks(ctx):
    assert PC.check_decompress_golden(ctx, "p224") > 40


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_sign_golden(ctx, curve):
    assert PC.check_sign_golden(ctx, curve) >= 12


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
    assert PC.check_codec_random(ctx, curve, n=600) > 0


def test_der_fuzz(ctx):
    assert PC.check_der_fuzz(ctx) > 2000


def test_recover_p224_tonelli_shanks(ctx):
    """p224 (p = 1 mod 4) recovers through the device's Tonelli-Shanks square root"""
    assert PC.check_recover_golden(ctx, "p224") > 20


def test_eddsa_sign_golden(ctx):
    assert PC.check_eddsa_sign_golden(ctx) > 100


def test_eddsa_verify_golden(ctx):
    assert PC.check_eddsa_golden(ctx) > 200


@pytest.mark.parametrize("n", [1, 7, 8, 9, 13, 41, 203])
def test_host_pipeline_chunks(ctx, n):
    """The host-buffer entry points cut a batch into chunks on two alternating lanes (the
    hostsim quantum is 8 items, so these sizes give 1..7 chunks incl. an absorbed tail);
    results must not depend on the cut.  Checked against the C oracle item by item."""
    from oracle import c_oracle
    rnd = np.random.default_rng(1234 + n)
    d = rnd.integers(0, 256, (n, 32), dtype=np.uint8)
    k = rnd.integers(0, 256, (n, 32), dtype=np.uint8)
    k2 = rnd.integers(0, 256, (n, 32), dtype=np.uint8)
    k[0] = 0                                     # infinity in the first chunk
    pts, inf = c_oracle.mul("secp256k1", d)
    assert not inf.any()
    want = c_oracle.mul("secp256k1", k, pts)
    got = ctx.mul_var("secp256k1", k, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    bufs = (np.full((n, 64), 0xAA, np.uint8), np.full(n, 0xAA, np.uint8))
    got = ctx.mul_var("secp256k1", k, pts, out=bufs)
    assert got[0] is bufs[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    want = c_oracle.mul_add("secp256k1", k, None, k2, pts)
    got = ctx.mul_add2("secp256k1", k, None, k2, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    want = c_oracle.mul_add("secp256k1", k, pts[::-1].copy(), k2, pts)
    got = ctx.mul_add2("secp256k1", k, pts[::-1].copy(), k2, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    # verify: valid signatures built from the oracle's own k*G, every third one corrupted
    N = int(O.get_curve("secp256k1", False).n)
    ds = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in d]
    ks = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in k2]
    ss = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in rnd.integers(0, 256, (n, 32), dtype=np.uint8)]
    pub, _ = c_oracle.mul("secp256k1", elliptic_amd.ints_to_be(ds, 32))
    R, _ = c_oracle.mul("secp256k1", elliptic_amd.ints_to_be(ks, 32))
    rs = [int.from_bytes(R[i, :32].tobytes(), "big") % N for i in range(n)]
    zs = [(ss[i] * ks[i] - rs[i] * ds[i]) % N for i in range(n)]
    h, r, s_ = (elliptic_amd.ints_to_be(v, 32) for v in (zs, rs, ss))
    h[::3, 31] ^= 1
    want = c_oracle.verify("secp256k1", h, r, s_, pub)
    got = ctx.ecdsa_verify("secp256k1", h, r, s_, pub)
    assert np.array_equal(got, want)
    assert want[1::3].all() and not want[::3].any()


@pytest.mark.parametrize("curve", O.SHORT_CURVES)
def test_random_ladders_vs_c_oracle(ctx, curve):
    """seeded random and edge scalars through mul_var / mul_fixed / mul_add2 against the C port of
    the reference's ladder (a second, independent check of the odd-digit / affine-table paths)"""
    from oracle import c_oracle
    B = elliptic_amd.FIELD_BYTES[curve]
    cur = O.get_curve(curve, False)
    n_int = int(cur.n)
    rnd = np.random.default_rng(2025 + B)
    m = 40
    d = rnd.integers(0, 256, (m, B), dtype=np.uint8)
    k = rnd.integers(0, 256, (m, B), dtype=np.uint8)
    k2 = rnd.integers(0, 256, (m, B), dtype=np.uint8)
    if curve == "p521":
        d[:, 0] &= 1
        k[:, 0] &= 1
        k2[:, 0] &= 1
    edge = [0, 1, 2, 3, 15, 16, 17, n_int - 1, n_int, n_int + 1, (1 << (8 * B - (7 if curve == "p521" else 0))) - 1,
            n_int - 2, (n_int + 1) // 2, 1 << 4, (1 << 64) - 1, 1 << 128]
    for i, v in enumerate(edge):
        k[i] = np.frombuffer(int(v).to_bytes(B, "big"), np.uint8)
    pts, inf = c_oracle.mul(curve, d)
    assert not inf.any()
    want = c_oracle.mul(curve, k, pts)
    got = ctx.mul_var(curve, k, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    want = c_oracle.mul(curve, k)
    got = ctx.mul_fixed(curve, k)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    want = c_oracle.mul_add(curve, k, None, k2, pts)
    got = ctx.mul_add2(curve, k, None, k2, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])
    want = c_oracle.mul_add(curve, k, pts[::-1].copy(), k2, pts)
    got = ctx.mul_add2(curve, k, pts[::-1].copy(), k2, pts)
    assert np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0])


def test_empty_batches_newer_entry_points(ctx):
    """n = 0 through sign (both nonce sources), recover, decompress, EdDSA sign / verify"""
    z = np.zeros((0, 32), np.uint8)
    e32 = np.zeros((0, 32), np.uint8)
    r, s_, rec, ok = ctx.ecdsa_sign("secp256k1", z, e32, e32)
    assert r.shape == (0, 32) and ok.shape == (0,)
    r, s_, rec, ok = ctx.ecdsa_sign_det("secp256k1", z, e32)
    assert r.shape == (0, 32) and ok.shape == (0,)
    xy, st = ctx.ecdsa_recover("secp256k1", z, e32, e32, np.zeros(0, np.uint8))
    assert xy.shape == (0, 64) and st.shape == (0,)
    xy, ok = ctx.decompress("secp256k1", e32, np.zeros(0, np.uint8))
    assert xy.shape == (0, 64)
    sig, pub = ctx.eddsa_sign([], np.zeros((0, 32), np.uint8))
    assert sig.shape == (0, 64) and pub.shape == (0, 32)
    ok, err = ctx.eddsa_verify([], np.zeros((0, 64), np.uint8), np.zeros((0, 32), np.uint8))
    assert ok.shape == (0,)
    # codecs, validation, DER, wire verify, point addition
    xy, st = ctx.decode_points("secp256k1", np.zeros((0, 33), np.uint8))
    assert xy.shape == (0, 64) and st.shape == (0,)
    assert ctx.encode_points("secp256k1", np.zeros((0, 64), np.uint8), compact=True).shape == (0, 33)
    assert ctx.validate("ed25519", np.zeros((0, 64), np.uint8)).shape == (0,)
    r, s_, st = ctx.sig_from_der("p256", [])
    assert r.shape == (0, 32) and st.shape == (0,)
    assert ctx.sig_to_der("p256", e32, e32) == []
    ok, err = ctx.ecdsa_verify_wire("secp256k1", z, [], np.zeros((0, 33), np.uint8))
    assert ok.shape == (0,) and err.shape == (0,)
    xy, inf = ctx.point_add("p384", np.zeros((0, 96), np.uint8), np.zeros((0, 96), np.uint8))
    assert xy.shape == (0, 96)
    # argument errors are reported, not computed
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.decode_points("ed25519", np.zeros((1, 33), np.uint8))
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.sig_from_der("ed25519", [b"\x30\x00"])
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.point_add("curve25519", np.zeros((1, 64), np.uint8), np.zeros((1, 64), np.uint8))


def test_error_paths(ctx, hs):
    with pytest.raises(elliptic_amd.EllgpuError):
        ctx.mul_fixed("curve25519", np.zeros((1, 32), np.uint8))
    with pytest.raises(elliptic_amd.EllgpuError) as e:
        ctx.mul_add2("curve25519", np.zeros((1, 32), np.uint8), None, np.zeros((1, 32), np.uint8),
                     np.zeros((1, 64), np.uint8))
    assert "Not supported on Montgomery curve" in str(e.value)      # mont.js:155-157
    assert hs.ellgpu_curve_id(b"secp256k1") == 0 and hs.ellgpu_curve_id(b"nope") == -1
    # empty batch is a no-op
    out, inf = ctx.mul_fixed("secp256k1", np.zeros((0, 32), np.uint8))
    assert out.shape == (0, 64)


def test_fixed_base_table_narrows_when_memory_is_short(hs, monkeypatch):
    """ADVICE r3 / VERDICT r4 #7: the 256-bit curves' signed comb carries its window width in front
    of its first entry, so a context that cannot allocate the default table builds a narrower one
    and the SAME kernels run on it -- instead of ELLGPU_E_NOMEM on the first G*k / verify / sign.
    ELLGPU_COMB_MAX_BYTES makes allocations above a size fail (here: the host build's 8-bit comb
    of 33 x 128 entries does not fit, the 4-bit one of 65 x 8 does)."""
    c0 = elliptic_amd.Context(0, lib_path=hs)
    assert c0.comb_bits("secp256k1") == 0
    c0.reserve("secp256k1", 16)
    wide = c0.comb_bits("secp256k1")
    assert wide == 8                                      # tests/hostsim/build.py: -DELL_COMB_BITS_256=8
    c = _fresh_ctx(hs, monkeypatch, ELLGPU_COMB_MAX_BYTES=str(100 * 1024))
    for curve in ("secp256k1", "p256"):
        c.reserve(curve, 16)
        assert c.comb_bits(curve) == 4, curve
        assert PC.check_mul_golden(c, curve) > 50         # fixed-base rows included
        assert PC.check_verify_golden(c, curve) > 15
        assert PC.check_sign_golden(c, curve) > 10
        assert PC.check_signdet_golden(c, curve) > 10
    assert PC.check_recover_golden(c, "secp256k1") >= 30
    assert PC.check_exceptional_keys(c, "secp256k1") > 400
    rng = random.Random(3)
    ks = np.frombuffer(bytes(rng.getrandbits(8) for _ in range(40 * 32)), np.uint8).reshape(40, 32).copy()
    ks[0] = 0
    ks[1] = 255                                           # 2^256 - 1: every window at its maximum, the carry window used
    a, ai = c.mul_fixed("secp256k1", ks)
    b, bi = c0.mul_fixed("secp256k1", ks)
    assert np.array_equal(a, b) and np.array_equal(ai, bi)
    # nothing fits: the call fails loudly, and says why
    c2 = _fresh_ctx(hs, monkeypatch, ELLGPU_COMB_MAX_BYTES="64")
    with pytest.raises(elliptic_amd.EllgpuError) as e:
        c2.mul_fixed("secp256k1", ks)
    assert "comb table allocation failed" in str(e.value)
    for x in (c0, c, c2):
        x.close()


def test_one_context_entered_from_several_threads(ctx):
    """The N-API addon runs Promise-form batches on a libuv worker while the JS thread makes
    synchronous calls on the SAME context: entry points of one context take turns (a
    mutex in the C ABI), so concurrent callers get their own results, not each other's staged
    inputs.  ctypes releases the GIL around the calls: the threads really are inside together."""
    import threading
    from golden_util import I, verify_cases
    from elliptic_amd import ints_to_be
    cs = [c for c in verify_cases("secp256k1") if len(c["z"]) == 64]
    h = ints_to_be([I(c["z"]) for c in cs], 32)
    r = ints_to_be([I(c["r"]) for c in cs], 32)
    s = ints_to_be([I(c["s"]) for c in cs], 32)
    q = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], 32), ints_to_be([I(c["qy"]) for c in cs], 32)], axis=1)
    want = np.array([1 if c["ok"] else 0 for c in cs], np.uint8)
    rng = random.Random(11)
    ks = np.frombuffer(bytes(rng.getrandbits(8) for _ in range(24 * 32)), np.uint8).reshape(24, 32)
    want_xy, want_inf = ctx.mul_fixed("secp256k1", ks)
    errors = []

    def verifier(t):
        try:
            for it in range(6):
                lo = (t * 7 + it * 3) % (len(cs) - 8)
                m = 8 + (t + it) % 5
                got = ctx.ecdsa_verify("secp256k1", h[lo:lo + m], r[lo:lo + m], s[lo:lo + m], q[lo:lo + m])
                if not np.array_equal(got, want[lo:lo + m]):
                    errors.append(("verify", t, it))
        except Exception as e:                      # noqa: BLE001
            errors.append(("verify raised", t, repr(e)))

    def multiplier(t):
        try:
            for it in range(6):
                lo = (t + it) % 12
                xy, inf = ctx.mul_fixed("secp256k1", ks[lo:lo + 12])
                if not (np.array_equal(xy, want_xy[lo:lo + 12]) and np.array_equal(inf, want_inf[lo:lo + 12])):
                    errors.append(("mul_fixed", t, it))
        except Exception as e:                      # noqa: BLE001
            errors.append(("mul raised", t, repr(e)))

    th = [threading.Thread(target=verifier if t % 2 == 0 else multiplier, args=(t,)) for t in range(6)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errors, errors[:5]


def test_point_operands_and_results_must_not_overlap(ctx, hs):
    """*_dev entry points read the operands again after the results are written (the curve test):
    an out_xy over its own in_xy is refused, loudly"""
    n = 4
    buf = np.zeros((n, 64), np.uint8)
    k = np.zeros((n, 32), np.uint8)
    inf = np.zeros(n, np.uint8)
    rc = hs.ellgpu_mul_var_dev(ctx._ctx, 0, n, k.ctypes.data, buf.ctypes.data, buf.ctypes.data, inf.ctypes.data, None)
    assert rc == -2 and b"overlap" in hs.ellgpu_last_error()
    rc = hs.ellgpu_mul_add2_dev(ctx._ctx, 0, n, k.ctypes.data, None, k.ctypes.data, buf.ctypes.data,
                                buf[1:].ctypes.data, inf.ctypes.data, None)
    assert rc == -2 and b"overlap" in hs.ellgpu_last_error()


def test_device_group_shards_match_single_context():
    """ellgpu_group_create: the sharded host entry points (one host thread per member, results
    written straight into the caller's buffers) give exactly the single-context results, for
    uneven shard sizes and for batches smaller than the group (CPU build of the
    same C ABI code, group members = two contexts of the simulated device)"""
    lib = _lib.load(build_hostsim(), optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing", "ellgpu_ctx_get_timing",
                                               "ellgpu_debug_field_op"))
    one = elliptic_amd.Context(0, lib_path=lib)
    grp = elliptic_amd.Context(lib_path=lib, devices=[0, 0, 0])
    assert one.group_size() == 1 and grp.group_size() == 3
    rnd = np.random.RandomState(7)
    for curve, n in (("secp256k1", 101), ("p256", 2), ("ed25519", 37), ("secp256k1", 1)):
        B = elliptic_amd.FIELD_BYTES[curve]
        k = rnd.randint(0, 256, (n, B)).astype(np.uint8)
        d = rnd.randint(0, 256, (n, B)).astype(np.uint8)
        if curve == "ed25519":
            k[:, 0] &= 0x0F
            d[:, 0] &= 0x0F
        p1, i1 = one.mul_fixed(curve, d)
        pg, ig = grp.mul_fixed(curve, d)
        assert np.array_equal(p1, pg) and np.array_equal(i1, ig)
        a1, b1 = one.mul_var(curve, k, p1)
        ag, bg = grp.mul_var(curve, k, p1)
        assert np.array_equal(a1, ag) and np.array_equal(b1, bg)
        a1, b1 = one.mul_add2(curve, k, None, d, p1)
        ag, bg = grp.mul_add2(curve, k, None, d, p1)
        assert np.array_equal(a1, ag) and np.array_equal(b1, bg)
    # verify: golden tuples through the group
    from golden_util import I, verify_cases
    cs = [c for c in verify_cases("secp256k1") if len(c["z"]) == 64]
    from elliptic_amd import ints_to_be
    h = ints_to_be([I(c["z"]) for c in cs], 32)
    r = ints_to_be([I(c["r"]) for c in cs], 32)
    s = ints_to_be([I(c["s"]) for c in cs], 32)
    q = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], 32), ints_to_be([I(c["qy"]) for c in cs], 32)], axis=1)
    ok = grp.ecdsa_verify("secp256k1", h, r, s, q)
    assert [bool(x) for x in ok] == [c["ok"] for c in cs]
    # an error inside a shard surfaces with the shard's message
    with pytest.raises(elliptic_amd.EllgpuError):
        grp.mul_var("curve25519", np.zeros((4, 32), np.uint8), np.zeros((4, 64), np.uint8))
    # reserve / synchronize fan out to every member; null buffers are refused before sharding
    grp.reserve("secp256k1", 1000)
    grp.synchronize()
    import ctypes
    assert lib.ellgpu_mul_fixed(grp._ctx, 0, 4, None, None, None) == -2
    assert lib.ellgpu_ecdsa_verify(grp._ctx, 0, 4, None, 32, 0, None, None, None, None, None) == -2
    # a user-defined curve is registered with all members or with none: fill ONE member's table
    # behind the group's back, then the group definition must fail and leave member 0 untouched
    members = grp.group_size()
    p256 = O.get_curve("p256", False)
    cid = grp.define_short(p256.p, 5, 7)
    assert cid >= 16 and grp.define_short(p256.p, 5, 7) == cid          # same parameters, same id
    solo = elliptic_amd.Context(0, lib_path=lib)
    ids = {solo.define_short(p256.p, 5, 100 + j) for j in range(16)}
    assert len(ids) == 16
    with pytest.raises(elliptic_amd.EllgpuError):
        solo.define_short(p256.p, 5, 999)                               # the seventeenth
    solo.close()
    assert members == 3
    one.close()
    grp.close()


def test_secp256k1_on_the_lazy_field():
    """-DELL_K256_LAZY=1 (secp256k1 on the 9 x 29-bit signed-limb field, fpk256l.h: lazy
    additions, two-product multiplies, the libsecp256k1-style doubling) is a build switch of the
    product: its whole secp256k1 path against the reference's goldens, with the run-time column
    bound checks of the host build (ELL_BOUNDS_CHECK) armed"""
    lib = _lib.load(build_hostsim(lazy_k256=True), optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing",
                                                              "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
    c = elliptic_amd.Context(0, lib_path=lib)
    assert PC.check_mul_golden(c, "secp256k1") > 100
    assert PC.check_verify_golden(c, "secp256k1") > 20
    assert PC.check_sign_golden(c, "secp256k1") >= 10
    assert PC.check_recover_golden(c, "secp256k1") >= 30
    assert PC.check_decompress_golden(c, "secp256k1") >= 20
    assert PC.check_add_golden(c, "secp256k1") >= 10
    c.close()


@pytest.mark.parametrize("field", [2, 3])
def test_lazy_field_two_product_and_half(hs, field):
    """FpK256L's lazy forms (field id 2) and the same algebra across a 16-lane row (FpK256C, field
    id 3): x*y + (4p - x)*(x - y + 4p) through mul2 and 3/2 x^2 through half_l, edge and random
    operands, incl. p - 1 (whose products fold to a slightly NEGATIVE value: the canonicalisation
    must cope)"""
    p = FIELDS[field]
    rnd = random.Random(77)
    edge = [0, 1, 2, p - 1, p - 2, p, p + 1, 2 ** 256 - 1, 2 ** 255, 2 ** 232, 2 ** 232 - 1, 2 ** 29, 2 ** 29 - 1, (p + 1) // 2]
    vals = [(x, y) for x in edge for y in edge] + [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(400)]
    inv2 = pow(2, -1, p)
    for x, y in vals:
        for op, want in ((11, (2 * x * y - x * x) % p), (12, 3 * x * x * inv2 % p), (2, x * y % p), (1, (x - y) % p)):
            r = (ctypes.c_uint32 * 8)()
            assert hs.hs_field_op(field, op, _limbs(x, 8), _limbs(y, 8), r) == 0
            assert _val(r) == want, (op, hex(x), hex(y))


@pytest.mark.parametrize("field", [13, 14])
def test_solinas_pair_products(hs, field):
    """FpSolinas::mul_sub_mul / mul_sub_sqr8 (p256, p384): a difference of two wide products through
    ONE signed lazy fold -- negative differences, operands at the edges of [0, p), the rare top
    branch -- against Python integers"""
    p = FIELDS[field]
    L = hs.hs_field_limbs(field)
    rnd = random.Random(4242 + field)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 2 ** 32 - 1, 2 ** (32 * (L - 1)), p - 2 ** 32, p - 2 ** 96]
    vals = [(x, y) for x in edge for y in edge] + [(rnd.randrange(p), rnd.randrange(p)) for _ in range(600)]
    for x, y in vals:
        for op, want in ((17, (x * y - (y - x) * (x + y)) % p), (18, (x * y - 8 * (x - y) ** 2) % p)):
            r = (ctypes.c_uint32 * L)()
            assert hs.hs_field_op(field, op, _limbs(x, L), _limbs(y, L), r) == 0
            assert _val(r) == want, (field, op, hex(x), hex(y))


def test_row_field_corners(hs):
    """FpK256C (csrc/coop.h) on the host simulation of its 16-lane row: the exact zero test of a
    product (factors whose product is 0 or p exactly, and near misses), a lazy difference times a
    lazy sum, entries read from the one-lane tables' memory format with and without the lazy
    negation -- against Python integers"""
    p = FIELDS[3]
    rnd = random.Random(1234)
    import math
    rt = math.isqrt(p)
    pairs = [(0, 5), (5, 0), (1, p), (p, 1), (p, p), (2, (p + 1) // 2), (p - 1, p - 1), (1, 0), (rt, rt), (rt + 1, rt + 1),
             (1 << 29, 1 << 227), (1 << 128, 1 << 128), ((1 << 256) - 1, (1 << 256) - 1), (977, 1 << 29), (p - 977, 3)]
    # products that are small multiples of 2^29 (limb 0 of the unreduced product is zero)
    pairs += [(1 << 29, k) for k in (1, 2, 3, 1 << 29, (1 << 58) + 1)] + [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(300)]
    for x, y in pairs:
        for op, want in ((14, x * y % p), (15, (x - y) * (x + y) % p), (16, (-x if y & 1 else x) % p)):
            r = (ctypes.c_uint32 * 8)()
            assert hs.hs_field_op(3, op, _limbs(x, 8), _limbs(y, 8), r) == 0
            assert _val(r) == want, (op, hex(x), hex(y))


@pytest.mark.parametrize("idx", range(5))
def test_user_defined_short_curves(ctx, idx):
    """`new curve.short({p, a, b})` with parameters that are no preset: run-time modulus, generic-a
    doubling (CvCustom / FpMontRT) against the reference's results"""
    spec = PC.custom_curves()[idx]
    assert PC.check_custom_short_golden(ctx, spec) > 80


def test_user_defined_curve_errors(ctx):
    import elliptic_amd
    with pytest.raises(elliptic_amd._lib.EllgpuError):
        ctx.define_short(2 ** 200, 1, 1)                     # even modulus
    with pytest.raises(elliptic_amd._lib.EllgpuError):
        ctx.define_short(3, 1, 1)
    cid = ctx.define_short(PC.I(PC.custom_curves()[2]["p"]), 5, 7)
    k = np.zeros((1, 32), np.uint8)
    with pytest.raises(elliptic_amd._lib.EllgpuError) as e:
        ctx.mul_fixed(cid, k)
    assert e.value.code == -5                                # ELLGPU_E_UNSUPPORTED
    with pytest.raises(elliptic_amd._lib.EllgpuError):
        ctx.mul_var(31, k, np.zeros((1, 64), np.uint8))        # an id nobody defined


@pytest.mark.parametrize("idx", range(4))
def test_user_defined_edwards_curves(ctx, idx):
    """`new curve.edwards({p, a, c: 1, d})` with parameters that are not ed25519's: projective
    (a != -1) and extended (a = -1) curves of the reference against the device's run-time-prime
    projective ladder (EdcWork)"""
    assert PC.check_custom_edwards_golden(ctx, PC.custom_edwards_curves()[idx]) > 80


def test_deferred_small_calls(ctx):
    """the split form of a few-item call (ellgpu_ctx_defer / ellgpu_ctx_collect), CPU build"""
    PC.check_deferred_calls(ctx)


def test_x25519_derive(ctx):
    """KeyPair#derive on curve25519 as one call: validate (Euler's criterion) beside the ladder"""
    assert PC.check_x25519_derive(ctx, count=40) >= 50
