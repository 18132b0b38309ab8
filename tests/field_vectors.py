"""Directed operand pairs for the rarely taken branches of the special-form fields
(FpK256 = field 0, Fp25519 = field 1 of the field-op probes): carry / borrow ripples that need a
limb to be exactly 2^32 - 1 or 0, and values that land in [p, 2^256) before the last
subtraction.  Random operands reach these with probability ~2^-31, so they are constructed.
Python mirrors of the two reduce_wide routines assert that the multiply vectors really enter
the branch."""

M = 1 << 256
P_K256 = M - (1 << 32) - 977
P_25519 = (1 << 255) - 19


def k256_mul_trace(a, b):
    """mirror of FpK256::reduce_wide's intermediates -> (carry out of limb 2, u[7])"""
    N = a * b
    lo, hi = N % M, N // M
    delta = (1 << 32) + 977
    u = lo + hi * delta                                # 10 limbs
    T, ulow = u >> 256, u % M
    c = ((ulow % (1 << 96)) + T * delta) >> 96
    return c, (ulow >> 224) & 0xFFFFFFFF


def f25519_mul_trace(a, b):
    """mirror of Fp25519::reduce_wide -> (carry out of limb 0 in finish, u[7] & 0x7fffffff)"""
    N = a * b
    lo, hi = N % M, N // M
    u = lo + 38 * hi
    carry, ul = u >> 256, u % M
    top = ul >> 255
    ul &= (1 << 255) - 1
    add0 = carry * 38 + 19 * top
    return ((ul & 0xFFFFFFFF) + add0) >> 32, (ul >> 224) & 0x7FFFFFFF


def k256_fold_overflow(a, b):
    """mirror of the device FpK256::reduce_wide: does any x_i = hi_i*977 + lo_i + hi_i*2^32
    overflow 64 bits (the wave then takes the generic fold)?"""
    N = a * b
    for i in range(8):
        lo = (N >> (32 * i)) & 0xFFFFFFFF
        hi = (N >> (32 * (8 + i))) & 0xFFFFFFFF
        if hi * 977 + lo + (hi << 32) >= 1 << 64:
            return True
    return False


def fold_vectors():
    """-> (field 0, op 2) vectors around the overflow boundary of the device fold: high product
    limbs in [2^32 - 980, 2^32), both overflowing and not, mixed so that lanes of one wave
    disagree"""
    import random
    rnd = random.Random(2024)
    p = P_K256
    out = []
    n_over = n_not = 0
    for trial in range(4000):
        j = rnd.randrange(1, 8)
        a = rnd.getrandbits(256)
        # force one limb of a (it becomes a high limb of a * 2^(32 j)(+1)) to the boundary region
        pos = rnd.randrange(8 - j, 8)
        limb = 0xFFFFFFFF - rnd.randrange(0, 985)
        a = (a & ~(0xFFFFFFFF << (32 * pos))) | (limb << (32 * pos))
        a %= p
        b = ((1 << (32 * j)) + rnd.choice([0, 1, 1, 0xFFFFFFFF, 977, rnd.getrandbits(32)])) % p
        over = k256_fold_overflow(a, b)
        if over and n_over < 150:
            n_over += 1
        elif not over and n_not < 150:
            n_not += 1
        else:
            continue
        out.append((0, 2, a, b, a * b % p))
        if n_over >= 150 and n_not >= 150:
            break
    assert n_over >= 50 and n_not >= 50, (n_over, n_not)
    rnd.shuffle(out)
    return out


def f25519_fold_overflow(a, b):
    """mirror of the device Fp25519::reduce_wide: does 38*hi_2j + (lo_2j, lo_2j+1) overflow?"""
    N = a * b
    for j in range(4):
        lo = (N >> (64 * j)) & ((1 << 64) - 1)
        hi = (N >> (32 * (8 + 2 * j))) & 0xFFFFFFFF
        if 38 * hi + lo >= 1 << 64:
            return True
    return False


def fold25519_vectors():
    """(field 1, op 2) vectors on both sides of the device fold's overflow boundary: a low
    64-bit pair within 38 * 2^32 of 2^64"""
    import random
    rnd = random.Random(777)
    p = P_25519
    out = []
    n_over = n_not = 0
    for trial in range(20000):
        # N = a * b with b = 2^(64 j) + small: low pairs of N are pairs of a
        j = rnd.randrange(0, 3)
        a = rnd.getrandbits(255)
        pos = rnd.randrange(0, 4 - j)
        pair = (1 << 64) - 1 - rnd.randrange(0, 40 << 32)
        a = (a & ~(((1 << 64) - 1) << (64 * pos))) | (pair << (64 * pos))
        a %= p
        b = ((1 << (64 * j)) + (rnd.getrandbits(30) << 64 * 3)) % p if j else (1 + (rnd.getrandbits(60) << 190)) % p
        over = f25519_fold_overflow(a, b)
        if over and n_over < 120:
            n_over += 1
        elif not over and n_not < 120:
            n_not += 1
        else:
            continue
        out.append((1, 2, a, b, a * b % p))
        if n_over >= 120 and n_not >= 120:
            break
    assert n_over >= 30 and n_not >= 30, (n_over, n_not)
    rnd.shuffle(out)
    return out


def rare_vectors():
    """-> list of (field, op, a, b, expected); op: 0 add, 1 sub, 2 mul, 3 sqr (b ignored)"""
    out = []

    def put(field, p, op, a, b):
        assert 0 <= a < p and 0 <= b < p, (field, op, hex(a), hex(b))
        want = [(a + b) % p, (a - b) % p, a * b % p, a * a % p][op]
        out.append((field, op, a, b, want))

    # ---- secp256k1 field ----
    p = P_K256
    delta = M - p
    for t in [(0xFFFFFFFF << 32) | (2 ** 32 - 977), (0xFFFFFFFF << 32) | 0xFFFFFFFF, (0xFFFFFFFE << 32) | 0xFFFFFFFF,
              (2 ** 224 - 1), (2 ** 224 - 1) - 976, ((2 ** 192 - 1) << 32) | 0xFFFFFC2F, (1 << 64) - 977, (1 << 64) - 978]:
        S = M + t                                      # a + b = 2^256 + t: the fold ripples out of limb 1
        a = p - 1 - 12345
        put(0, p, 0, a, S - a)
        put(0, p, 0, S - a, a)
    for S in (p, p + 1, M - 1, p - 1, M - 2 ** 224, M - 2 ** 224 - 1):   # carry-less sums around [p, 2^256)
        put(0, p, 0, S // 2, S - S // 2)
        put(0, p, 0, S - S // 2, S // 2)
    for t in [delta + 1, delta + 2, (1 << 33) + 976, (1 << 33) + 977, (1 << 33) + 978, (1 << 64) + 5,
              (1 << 224) + 976, (2 << 32) + 3, (1 << 64) + (1 << 32) + 976, (1 << 96) + (1 << 32) + 5]:
        # a - b = t - 2^256 (borrow); t - delta then borrows out of limb 1 when limbs 0..1 of t are small
        b = p - 1
        put(0, p, 1, (t - M) + b, b)
    for a, b in ((0, p - 1), (0, 1), (5, 5), (0, 0), (1, 2), (976, 977), (p - 1, p - 1)):
        put(0, p, 1, a, b)
    # mul: carry out of limb 2 in the second fold.  b = 2^255: N = a << 255, hi = a >> 1; choose
    # x = a >> 1 with x*delta = -1-k (mod 2^96) and x large so that T >= 1
    inv = pow(delta, -1, 1 << 96)
    hits = 0
    for k in range(40):
        x = ((1 << 96) - 1 - k) * inv % (1 << 96) + ((0x7ACE << 238) | (k << 100))
        a = 2 * x
        if a >= p:
            continue
        hits += k256_mul_trace(a, 1 << 255)[0]
        put(0, p, 2, a, 1 << 255)
        put(0, p, 2, 1 << 255, a)
    assert hits >= 5, hits
    # u[7] = 2^32 - 1 and values in [p, 2^256) before the last subtraction
    for a in (p - 1, p - 2, M - 2 ** 224 + 5, p - 2 ** 200):
        assert k256_mul_trace(a, 1)[1] == 0xFFFFFFFF
        put(0, p, 2, a, 1)
        put(0, p, 3, a, 0)
        put(0, p, 2, a, p - 1)
    # ---- 2^255 - 19 ----
    p = P_25519
    for S in (p, p + 1, p + 18, (1 << 255) - 1, 1 << 255, (1 << 255) + 1, 2 * p - 2, (1 << 255) + (1 << 32) - 19,
              (1 << 255) + (1 << 32) - 20, (1 << 255) + (1 << 64) - 19, p - 1):
        put(1, p, 0, S // 2, S - S // 2)
        put(1, p, 0, S - S // 2, S // 2)
    for a, b in ((0, 1), (0, p - 1), (5, 6), (18, 19), (0, 19), (1 << 32, (1 << 32) + 1), (7, 7), (p - 1, 0),
                 (20, (1 << 64) + 39), (3, 1 << 200), (0, 0), (1 << 32, (1 << 32) + 19), (1 << 64, (1 << 64) + 1)):
        put(1, p, 1, a, b)
    hits = 0
    for k in range(64):
        a = ((((1 << 31) - 1) << 224) | (0xFFFFFFFF - k)) % p    # low limb near 2^32, top limb large
        for b in (2, 3, 1 << 31, (1 << 31) + 1, p - 1):
            hits += f25519_mul_trace(a, b)[0]
            put(1, p, 2, a, b)
            put(1, p, 2, b, a)
    assert hits >= 1, hits
    for a in (p - 1, p - 2, (1 << 255) - 20, (1 << 254) + 1, 1 << 254):
        put(1, p, 3, a, 0)
        put(1, p, 2, a, 2)
        put(1, p, 2, a, 1)
    assert f25519_mul_trace(p - 1, 1)[1] == 0x7FFFFFFF
    return out


def shift_vectors():
    """-> list of (field, op, a, b, expected) for mul_pow2 (op 6/7/8 = x2/x4/x8) and, on field 1,
    mul_u32 (op 10, b = the one-limb constant): operands whose shifted limbs are all ones or
    whose folded top bits carry out of the low limbs."""
    import random
    rnd = random.Random(99)
    out = []
    for field, p in ((0, P_K256), (1, P_25519)):
        bits = 256 if field == 0 else 255
        for K in (1, 2, 3):
            cands = [0, 1, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, p >> K, (p >> K) + 1, (p >> K) - 1,
                     (1 << (bits - K)) - 1, 1 << (bits - K), (1 << (bits - K)) + 1]
            for top in range(1 << K):
                for lo in [(1 << bits) - (1 << K),                         # all ones after the shift
                           ((1 << 64) - 1) & ~((1 << K) - 1),              # limbs 0..1 all ones
                           (((1 << 32) - 1) << 32) | ((1 << 32) - (1 << K)),
                           (((1 << 32) - 1) << 32) | ((1 << 32) - 977 * top - (1 << K)) % (1 << 32) & ~((1 << K) - 1),
                           (1 << 32) - (1 << K), ((1 << 32) - 19 * top) & ~((1 << K) - 1) & 0xFFFFFFFF,
                           ((1 << bits) - (1 << 224)) & ~((1 << K) - 1),   # top limb all ones
                           rnd.getrandbits(bits) & ~((1 << K) - 1)]:
                    a = (top << (bits - K)) | (lo >> K)
                    cands.append(a % p)
            cands += [rnd.randrange(p) for _ in range(50)]
            for a in cands:
                out.append((field, 5 + K, a, 0, (a << K) % p))
    p = P_25519
    k = 121666
    half = pow(k // 2, -1, 1 << 31)
    for j in range(24):
        a0 = ((1 << 31) - 1 - j) * half % (1 << 31)          # a0 * k = 2^32 - 2 - 2j (mod 2^32)
        for hi in (0, 1, (1 << 223) - 1, rnd.getrandbits(222), (1 << 222) + 12345):
            a = ((hi << 32) | a0) % p
            out.append((1, 10, a, k, a * k % p))
    for a in [0, 1, p - 1, p - 2, (1 << 255) - 20, (p - 1) // 2, ((1 << 255) - 19) // k, ((1 << 255) - 19) // k + 1,
              (1 << 254) - 1] + [rnd.randrange(p) for _ in range(60)]:
        for kk in (k, 1, 2, (1 << 20) - 1, 38):
            out.append((1, 10, a, kk, a * kk % p))
    return out


# ---- FpSolinas (p192 / p224 / p256 / p384 = fields 11..14): the lazy-accumulator fold ----------------
SOL_FIELDS = {
    11: (2 ** 192 - 2 ** 64 - 1, 6, [(2, 1), (0, 1)]),
    12: (2 ** 224 - 2 ** 96 + 1, 7, [(3, 1), (0, -1)]),
    13: (2 ** 256 - 2 ** 224 + 2 ** 192 + 2 ** 96 - 1, 8, [(7, 1), (6, -1), (3, -1), (0, 1)]),
    14: (2 ** 384 - 2 ** 128 - 2 ** 96 + 2 ** 32 - 1, 12, [(4, 1), (3, 1), (1, -1), (0, 1)]),
}


def solinas_trace(a, b, field):
    """mirror of FpSolinas::reduce_wide -> (c2 after the second fold pass, top word, carry c)"""
    p, L, fold = SOL_FIELDS[field]
    N = a * b
    A = [(N >> (32 * k)) & 0xFFFFFFFF for k in range(2 * L)]
    for k in range(2 * L - 1, L - 1, -1):
        v = A[k]
        for pos, sg in fold:
            A[k - L + pos] += sg * v
    r, c = [], 0
    for k in range(L):
        t = A[k] + c
        r.append(t & 0xFFFFFFFF)
        c = t >> 32
    c2 = 0
    for k in range(fold[0][0] + 1):
        e = sum(sg * c for pos, sg in fold if pos == k)
        t = r[k] + e + c2
        r[k] = t & 0xFFFFFFFF
        c2 = t >> 32
    return c2, r[L - 1], c


def solinas_vectors():
    """(field, op 2, a, b, a*b mod p) for operand pairs found by a seeded structured search
    (tests/solinas_rare_operands.json) that enter the rare branch of the fold: a carry out of
    the second pass, or a top word of 2^32 - 1"""
    import json
    import os
    d = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "solinas_rare_operands.json")))
    out = []
    for field in (11, 12, 13, 14):
        p = SOL_FIELDS[field][0]
        n_c2 = n_top = 0
        for ah, bh in d[str(field)]:
            a, b = int(ah, 16), int(bh, 16)
            c2, top, _ = solinas_trace(a, b, field)
            n_c2 += c2 != 0
            n_top += top == 0xFFFFFFFF
            out.append((field, 2, a, b, a * b % p))
            out.append((field, 2, b, a, a * b % p))
            out.append((field, 3, a, 0, a * a % p))
        assert n_c2 >= 5 and n_top >= 10, (field, n_c2, n_top)
    return out


def solinas_chain_trace(T, field):
    """mirror of FpSolinas::reduce_wide_chain (p192 / p224 / p384) on a 2L-word value T ->
    (list of chain indices whose carry / borrow wanted to travel past the extra word,
     top word of the candidate result, candidate result).  Chains are numbered in execution
    order: stage 1 additions, stage 1 subtractions, stage 2 additions, stage 2 subtractions."""
    p, L, fold = SOL_FIELDS[field]
    P = fold[0][0]
    W = 1 << 32

    def run(words, src, n, flagged, base):
        order = [f for f in fold if f[1] > 0] + [f for f in fold if f[1] < 0]
        for idx, (pos, sg) in enumerate(order):
            c = 0
            for k in range(n):
                x = words[pos + k] + sg * (src[k] + c)
                c = 1 if (x < 0 or x >= W) else 0
                words[pos + k] = x % W
            if pos + n < len(words):
                x = words[pos + n] + sg * c
                c = 1 if (x < 0 or x >= W) else 0
                words[pos + n] = x % W
            if c:
                flagged.append(base + idx)
    t = [(T >> (32 * k)) & 0xFFFFFFFF for k in range(2 * L)]
    v = t[:L] + [0] * (P + 1)
    flagged = []
    run(v, t[L:], L, flagged, 0)
    g = v[L:]
    lo = v[:L]
    run(lo, g, P + 1, flagged, len(fold))
    return flagged, lo[L - 1], sum(w << (32 * k) for k, w in enumerate(lo))


def solinas_chain_vectors():
    """(field, op 13, low half, high half, T mod p): raw 2L-word values for the chain fold of
    p192 / p224 / p384 -- words drawn mostly from {0, 1, 2^32 - 2, 2^32 - 1} so that every chain's
    carry / borrow ripple and the top-word test are entered (counted with the mirror), next to
    values that stop one short of them, and plain random ones."""
    import random
    rnd = random.Random(384)
    special = [0, 1, 2, 0xFFFFFFFE, 0xFFFFFFFF, 0x80000000, 0x7FFFFFFF]
    out = []
    for field in (11, 12, 14):
        p, L, fold = SOL_FIELDS[field]
        hits = {}
        n_top = n_clean = 0
        for trial in range(6000):
            mode = trial % 3
            ws = []
            for k in range(2 * L):
                if mode == 0 or rnd.random() < (0.75 if mode == 1 else 0.3):
                    ws.append(rnd.choice(special))
                else:
                    ws.append(rnd.getrandbits(32))
            if trial % 7 == 0:
                ws = [rnd.getrandbits(32) for _ in range(2 * L)]
            T = sum(w << (32 * k) for k, w in enumerate(ws))
            flagged, top, cand = solinas_chain_trace(T, field)
            keep = False
            for c in flagged:
                if hits.get(c, 0) < 12:
                    hits[c] = hits.get(c, 0) + 1
                    keep = True
            if not flagged and top == 0xFFFFFFFF and n_top < 25:
                n_top += 1
                keep = True
            if not flagged and top != 0xFFFFFFFF:
                assert cand == T % p, (field, hex(T))            # the common path is exact
                if n_clean < 60:
                    n_clean += 1
                    keep = True
            if keep:
                lo, hi = T % (1 << (32 * L)), T >> (32 * L)
                out.append((field, 13, lo, hi, T % p))
        # every chain but the very first (whose extra word is V's zero top word) must have been
        # driven into its ripple
        assert set(hits) == set(range(1, 2 * len(fold))), (field, hits)
        assert n_top >= 5, (field, n_top)
    return out


def solinas_addsub_vectors():
    """(field, op, a, b, expected) for the rarely taken branch of FpSolinas::add / sub (p192, p224,
    p256, p384): a folded carry / borrow arriving at a limb whose word of 2^(32L) mod p is zero, and
    a carry-less sum in [p, 2^(32L))"""
    out = []
    for field in (11, 12, 13, 14):
        p, L, fold = SOL_FIELDS[field]
        C = (1 << (32 * L)) - p
        assert C == sum(sg << (32 * pos) for pos, sg in fold)
        w = [(C >> (32 * k)) & 0xFFFFFFFF for k in range(L)]
        # limbs that are copied (mode 0 of FpSolinas::limb_mode): a carry arriving there is the rare case
        mode0 = [k for k in range(L) if w[k] == 0 and not (k > 0 and w[k - 1] >= 0x10000)]
        pairs = [(p - 1, 1), (p - 1, C - 1) if C > 1 else (p - 1, 2), (p - 5, 7), (p - 1, p - 1),
                 ((p + 1) // 2, (p + 1) // 2)]
        subs = [(0, p - 1), (0, 1), (1, 2), (C, C + 1)]
        for k in mode0:
            # a + b = 2^(32L) + (B^k - 1) + x: every limb below k is 2^32 - 1 after the chain reaches it
            t = (1 << (32 * k)) - 1
            b_ = (1 << (32 * L)) + t - (p - 1)
            if 0 <= b_ < p:
                pairs.append((p - 1, b_))
            t2 = (1 << (32 * k)) - C if k and (1 << (32 * k)) > C else None
            if t2 is not None:
                b2 = (1 << (32 * L)) + t2 + (C - 1) - (p - 1)       # t + C ends in k limbs of ones + carry
                if 0 <= b2 < p:
                    pairs.append((p - 1, b2))
            # a - b borrows and t's limbs below k are smaller than C's: the borrow reaches limb k
            tt = (1 << (32 * k)) + C - 1 if k else None
            if tt is not None:
                bb = (1 << (32 * L)) - tt
                if 0 < bb < p:
                    subs.append((0, bb))
                    subs.append((5, bb + 5) if bb + 5 < p else (0, bb))
        n_rare_add = n_rare_sub = 0
        for a, b in pairs:
            assert 0 <= a < p and 0 <= b < p, (field, hex(a), hex(b))
            n_rare_add += solinas_addsub_rare(field, a, b, False)
            out.append((field, 0, a, b, (a + b) % p))
            out.append((field, 0, b, a, (a + b) % p))
        for a, b in subs:
            assert 0 <= a < p and 0 <= b < p
            n_rare_sub += solinas_addsub_rare(field, a, b, True)
            out.append((field, 1, a, b, (a - b) % p))
            out.append((field, 5, b, 0, (-b) % p))
        assert n_rare_add >= 3 and n_rare_sub >= 1, (field, n_rare_add, n_rare_sub)
    return out


def solinas_addsub_rare(field, a, b, sub):
    """mirror of FpSolinas::add / sub: does (a, b) enter the rarely taken branch?"""
    p, L, fold = SOL_FIELDS[field]
    C = (1 << (32 * L)) - p
    w = [(C >> (32 * k)) & 0xFFFFFFFF for k in range(L)]
    full = (a - b) if sub else (a + b)
    c = 1 if (full < 0 or full >> (32 * L)) else 0
    t = full % (1 << (32 * L))
    tl = [(t >> (32 * k)) & 0xFFFFFFFF for k in range(L)]
    rare = (not sub) and tl[L - 1] == 0xFFFFFFFF
    cc = 0
    for k in range(L):
        if w[k] != 0:
            v = tl[k] - (w[k] if c else 0) - cc if sub else tl[k] + (w[k] if c else 0) + cc
        elif k > 0 and w[k - 1] >= 0x10000:
            v = tl[k] - cc if sub else tl[k] + cc
        else:
            rare = rare or cc != 0
            cc = 0
            continue
        cc = 1 if (v < 0 or v >> 32) else 0
    return bool(rare or cc)


def p521_addsub_vectors():
    """(15, op, a, b, expected) for the rare branches of FpP521::add / sub: the folded bit 521 that
    carries out of limb 0, a sum equal to p, a repaid borrow that ripples"""
    p = (1 << 521) - 1
    out = []
    adds = [(p - 1, (1 << 32) + 1), (p - 1, 1 << 32), (p - 1, 1), ((p + 1) // 2, (p - 1) // 2), (p - 1, p - 1),
            (p - (1 << 32), 1 << 32), (p - 1, 2), ((1 << 521) - (1 << 480), (1 << 480) - 1),
            ((1 << 520) + 0xFFFFFFFF, 1 << 520), ((1 << 520) + 0xFFFFFFFE, (1 << 520) + 1)]
    for a, b in adds:
        assert 0 <= a < p and 0 <= b < p
        out.append((15, 0, a, b, (a + b) % p))
        out.append((15, 0, b, a, (a + b) % p))
    subs = [(0, 1), (0, 1 << 32), (0, (1 << 64)), (5, (1 << 32) + 5), (0, p - 1), (1, 2), ((1 << 96), (1 << 96) + (1 << 32)),
            (0, (1 << 520)), ((1 << 300), (1 << 300) + (1 << 512))]
    for a, b in subs:
        assert 0 <= a < p and 0 <= b < p
        out.append((15, 1, a, b, (a - b) % p))
        out.append((15, 5, b, 0, (-b) % p))
    return out
