"""GPU unit tests of the field layer (-m gpu): every field the engine uses,
add/sub/mul/sqr/inv/neg on edge + random operands, against Python integers.
This is what catches a wrong carry in the inline-asm multiply blocks directly,
before it shows up as a wrong curve point."""
import os
import random
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import elliptic_amd  # noqa: E402
from oracle import ec_oracle as O  # noqa: E402

pytestmark = pytest.mark.gpu

FIELDS = {
    0: O.get_curve("secp256k1", False).p, 1: 2 ** 255 - 19, 2: O.get_curve("secp256k1", False).p,   # 2 = the 9 x 29-bit form (generated asm)
    10: O.get_curve("secp256k1", False).p, 11: O.get_curve("p192", False).p,
    12: O.get_curve("p224", False).p, 13: O.get_curve("p256", False).p,
    14: O.get_curve("p384", False).p, 15: O.get_curve("p521", False).p,
    20: O.get_curve("secp256k1", False).n, 21: O.get_curve("p192", False).n,
    22: O.get_curve("p224", False).n, 23: O.get_curve("p256", False).n,
    24: O.get_curve("p384", False).n, 25: O.get_curve("p521", False).n,
    26: O.get_curve("ed25519", False).n,
}


@pytest.fixture(scope="module")
def ctx():
    c = elliptic_amd.Context(0)
    yield c
    c.close()


def _pack(vals, L):
    out = np.zeros((len(vals), L), np.uint32)
    for i, v in enumerate(vals):
        for l in range(L):
            out[i, l] = (v >> (32 * l)) & 0xFFFFFFFF
    return out


def _unpack(arr):
    return [sum(int(x) << (32 * l) for l, x in enumerate(row)) for row in arr]


@pytest.mark.parametrize("field", sorted(FIELDS))
def test_field_ops_gpu(ctx, field):
    p = FIELDS[field]
    L = (p.bit_length() + 31) // 32
    rnd = random.Random(4242 + field)
    top = (1 << (32 * L)) - 1
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, p >> 1, (1 << 32) - 1, 1 << 32, (1 << 64) - 1,
            top % p, (top >> 1) % p, int("ffffffff00000000" * L, 16) % p, int("00000000ffffffff" * L, 16) % p]
    import math
    rare = [(2, (p + 1) // 2), (3, (p + 2) // 3), (math.isqrt(p) + 1, math.isqrt(p) + 1),
            (math.isqrt(p) + 1, math.isqrt(p) + 2), (p - 1, p - 1), (p - 1, 2),   # land in [p, 2^(32L))
            (1, p - 1), (p - 1, 1), (2, p - 1), (p - 2, 2), (p - 2, 3), (0, 1), (1, 2), (0, p - 1)]   # sums / differences at the wrap
    a = edge + [x for x, _ in rare] + [rnd.randrange(p) for _ in range(1500)]
    b = [edge[(i * 7 + 3) % len(edge)] for i in range(len(edge))] + [y for _, y in rare] + [rnd.randrange(p) for _ in range(1500)]
    n = len(a)
    A, B = _pack(a, L), _pack(b, L)
    for op, fn in ((0, lambda x, y: (x + y) % p), (1, lambda x, y: (x - y) % p), (2, lambda x, y: x * y % p),
                   (3, lambda x, y: x * x % p), (5, lambda x, y: (-x) % p)):
        R = np.zeros((n, L), np.uint32)
        rc = ctx._lib.ellgpu_debug_field_op(ctx._ctx, field, op, n, A.ctypes.data, B.ctypes.data, R.ctypes.data)
        assert rc == 0
        got = _unpack(R)
        for i in range(n):
            assert got[i] == fn(a[i], b[i]), (field, op, hex(a[i]), hex(b[i]), hex(got[i]))
    # inversion (division steps, csrc/safegcd.h): edge values, powers of two, long zero runs, random
    inv_in = [0, 1, 2, 3, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 30, (1 << 30) - 1, (1 << 60) + 1, p >> 1]
    inv_in += [(1 << k) % p for k in range(1, p.bit_length(), 29)]
    inv_in += [((rnd.randrange(p) >> k) << k) % p for k in (1, 17, 30, 31, 61, 90)]
    inv_in += a[:400]
    m = len(inv_in)
    R = np.zeros((m, L), np.uint32)
    A2 = _pack(inv_in, L)
    assert ctx._lib.ellgpu_debug_field_op(ctx._ctx, field, 4, m, A2.ctypes.data, A2.ctypes.data, R.ctypes.data) == 0
    got = _unpack(R)
    for i in range(m):
        assert got[i] == (pow(inv_in[i], -1, p) if inv_in[i] else 0), (field, "inv", hex(inv_in[i]))


def test_rare_branches_gpu(ctx):
    """the directed vectors of tests/field_vectors.py through the device code (asm products)"""
    import field_vectors
    vecs = field_vectors.rare_vectors() + field_vectors.shift_vectors() + field_vectors.fold_vectors() + field_vectors.fold25519_vectors()
    for field in (0, 1):
        for op in (0, 1, 2, 3, 6, 7, 8, 10):
            sel = [v for v in vecs if v[0] == field and v[1] == op]
            if not sel:
                continue
            A, B = _pack([v[2] for v in sel], 8), _pack([v[3] for v in sel], 8)
            R = np.zeros((len(sel), 8), np.uint32)
            assert ctx._lib.ellgpu_debug_field_op(ctx._ctx, field, op, len(sel), A.ctypes.data, B.ctypes.data,
                                                  R.ctypes.data) == 0
            got = _unpack(R)
            for v, g in zip(sel, got):
                assert g == v[4], (field, op, hex(v[2]), hex(v[3]), hex(g))


def test_rare_branches_solinas_gpu(ctx):
    import field_vectors
    vecs = field_vectors.solinas_vectors() + field_vectors.solinas_addsub_vectors() + field_vectors.p521_addsub_vectors()
    for field, L in ((11, 6), (12, 7), (13, 8), (14, 12), (15, 17)):
        for op in (0, 1, 2, 3, 5):
            sel = [v for v in vecs if v[0] == field and v[1] == op]
            if not sel:
                continue
            A, B = _pack([v[2] for v in sel], L), _pack([v[3] for v in sel], L)
            R = np.zeros((len(sel), L), np.uint32)
            assert ctx._lib.ellgpu_debug_field_op(ctx._ctx, field, op, len(sel), A.ctypes.data, B.ctypes.data,
                                                  R.ctypes.data) == 0
            for v, g in zip(sel, _unpack(R)):
                assert g == v[4], (field, op, hex(v[2]), hex(v[3]), hex(g))


def test_solinas_chain_fold_raw_words_gpu(ctx):
    """FpSolinas::reduce_wide_chain (p192 / p224 / p384) on raw 2L-word values (field-op probe 13):
    directed values for every chain's rarely taken ripple and for the top-word test (the wave then
    redoes the reduction with the lazy-accumulator fold), near misses, and 20 000 random 2L-word
    values per field, lanes of one wave disagreeing about the branch"""
    import field_vectors
    vecs = field_vectors.solinas_chain_vectors()
    rnd = random.Random(1384)
    for field, L in ((11, 6), (12, 7), (14, 12)):
        p = FIELDS[field]
        sel = [(v[2], v[3], v[4]) for v in vecs if v[0] == field]
        assert len(sel) > 80
        for _ in range(20000):
            T = rnd.getrandbits(64 * L)
            sel.append((T % (1 << (32 * L)), T >> (32 * L), T % p))
        rnd.shuffle(sel)
        A, B = _pack([v[0] for v in sel], L), _pack([v[1] for v in sel], L)
        R = np.zeros((len(sel), L), np.uint32)
        assert ctx._lib.ellgpu_debug_field_op(ctx._ctx, field, 13, len(sel), A.ctypes.data, B.ctypes.data,
                                              R.ctypes.data) == 0
        for v, g in zip(sel, _unpack(R)):
            assert g == v[2], (field, hex(v[0]), hex(v[1]), hex(g))


def test_lazy_field_asm_gpu(ctx):
    """field id 2 = FpK256L through the generated v_mad_i64_i32 column statements
    (csrc/k256l_asm.h): the two-product multiply and the 3/2 x^2 of the lazy doubling, 20 000 random
    + edge operand pairs against Python integers (mul / sqr / add / sub / inv / neg of the same
    field run in test_field_ops_gpu[2])"""
    p = FIELDS[2]
    rnd = random.Random(4711)
    edge = [0, 1, 2, p - 1, p - 2, p, p + 1, 2 ** 256 - 1, 2 ** 255, 2 ** 232, 2 ** 232 - 1, 2 ** 29, 2 ** 29 - 1, (p + 1) // 2]
    vals = [(x, y) for x in edge for y in edge] + [(rnd.getrandbits(256), rnd.getrandbits(256)) for _ in range(20000)]
    A, B = _pack([v[0] for v in vals], 8), _pack([v[1] for v in vals], 8)
    inv2 = pow(2, -1, p)
    for op, fn in ((11, lambda x, y: (2 * x * y - x * x) % p), (12, lambda x, y: 3 * x * x * inv2 % p)):
        R = np.zeros((len(vals), 8), np.uint32)
        assert ctx._lib.ellgpu_debug_field_op(ctx._ctx, 2, op, len(vals), A.ctypes.data, B.ctypes.data, R.ctypes.data) == 0
        for (x, y), g in zip(vals, _unpack(R)):
            assert g == fn(x, y), (op, hex(x), hex(y), hex(g))
