"""CPU oracle for the scalar-multiplication hot path of indutny/elliptic 6.6.1.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (elliptic_amd/, the
C ABI, the HIP kernels) may import, call or link anything in this directory;
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do, and
only as the checker.

This is a restatement, on Python integers, of the algorithms the reference
runs in lib/elliptic/curve/{base,short,edwards,mont}.js, lib/elliptic/utils.js
and lib/elliptic/ec/index.js (file:line cited per function, relative to
/root/reference).  It deliberately keeps the reference's structure -- the same
signed-digit recodings, the same fixed-base comb over "doubles", the same
interleaved wNAF/JSF ladder with the GLV split, the same Jacobian / extended /
x-only formulas -- so that it exercises the same exceptional branches.

Parity is PINNED: tests/test_oracle_golden.py checks every function here
against tests/golden/*.json, which tools/gen_golden.js produced by running the
reference bundle itself (and the reference's own mocha suite, captured at the
Point.mul / mulAdd / jmulAdd / ec.verify boundary) in the build container.
Field arithmetic (bn.js 4.11.9 `red*`) is replaced by Python `%` -- values are
canonical residues in both, which is all the reference's results depend on.
"""

from __future__ import annotations

import json
import os
from dataclasses import dataclass, field
from typing import List, Optional, Tuple

Affine = Optional[Tuple[int, int]]          # None = point at infinity
Jac = Tuple[int, int, int]                  # Z == 0 -> infinity


# --------------------------------------------------------------------------
# Scalar recoding -- lib/elliptic/utils.js
# --------------------------------------------------------------------------

def get_naf(num: int, w: int, bits: int) -> List[int]:
    """utils.js:15-44 getNAF (the reference's non-minimal variant for w>1)."""
    naf = [0] * (max(num.bit_length(), bits) + 1)
    ws = 1 << (w + 1)
    k = num
    for i in range(len(naf)):
        mod = k & (ws - 1)
        if k & 1:
            if mod > (ws >> 1) - 1:
                z = (ws >> 1) - mod
            else:
                z = mod
            k -= z
        else:
            z = 0
        naf[i] = z
        k >>= 1
    return naf


def get_jsf(k1: int, k2: int) -> List[List[int]]:
    """utils.js:47-101 getJSF (joint sparse form of two non-negative ints)."""
    jsf: List[List[int]] = [[], []]
    d1 = d2 = 0
    while k1 > -d1 or k2 > -d2:
        m14 = ((k1 & 3) + d1) & 3
        m24 = ((k2 & 3) + d2) & 3
        if m14 == 3:
            m14 = -1
        if m24 == 3:
            m24 = -1
        if (m14 & 1) == 0:
            u1 = 0
        else:
            m8 = ((k1 & 7) + d1) & 7
            u1 = -m14 if (m8 in (3, 5) and m24 == 2) else m14
        jsf[0].append(u1)
        if (m24 & 1) == 0:
            u2 = 0
        else:
            m8 = ((k2 & 7) + d2) & 7
            u2 = -m24 if (m8 in (3, 5) and m14 == 2) else m24
        jsf[1].append(u2)
        if 2 * d1 == u1 + 1:
            d1 = 1 - d1
        if 2 * d2 == u2 + 1:
            d2 = 1 - d2
        k1 >>= 1
        k2 >>= 1
    return jsf


def div_round(a: int, n: int) -> int:
    """bn.js divRound (dist/elliptic.js:6387): round-half-up of a/n, sign-aware."""
    q, r = divmod(abs(a), n)
    if 2 * r >= n:
        q += 1
    return q if a >= 0 else -q


# --------------------------------------------------------------------------
# Short Weierstrass -- lib/elliptic/curve/short.js + base.js
# --------------------------------------------------------------------------

@dataclass
class _Tables:
    """BasePoint.precomputed (base.js:312-327): naf wnd + doubles step."""
    naf_wnd: int
    naf_points: list
    doubles_step: int
    doubles_points: list


class ShortPoint:
    """Affine point (short.js:253-281) with optional precomputed tables."""

    __slots__ = ("curve", "x", "y", "pre", "_beta")

    def __init__(self, curve, x, y):
        self.curve = curve
        self.x = x
        self.y = y
        self.pre: Optional[_Tables] = None
        self._beta = None

    @property
    def inf(self):
        return self.x is None

    def xy(self) -> Affine:
        return None if self.inf else (self.x, self.y)

    def neg(self, precompute=False):
        """short.js:458-480 (negates the tables as well when asked)."""
        if self.inf:
            return self
        c = self.curve
        r = ShortPoint(c, self.x, (-self.y) % c.p)
        if precompute and self.pre is not None:
            r.pre = _Tables(self.pre.naf_wnd,
                            [q.neg() for q in self.pre.naf_points],
                            self.pre.doubles_step,
                            [q.neg() for q in self.pre.doubles_points])
        return r

    def eq(self, o):
        return self.inf == o.inf and (self.inf or (self.x == o.x and self.y == o.y))

    def add(self, o):
        """short.js:365-392 affine chord (1 inversion)."""
        c = self.curve
        p = c.p
        if self.inf:
            return o
        if o.inf:
            return self
        if self.eq(o):
            return self.dbl()
        if self.neg().eq(o):
            return c.point(None, None)
        if self.x == o.x:
            return c.point(None, None)
        cc = (self.y - o.y) % p
        if cc != 0:
            cc = cc * pow((self.x - o.x) % p, -1, p) % p
        nx = (cc * cc - self.x - o.x) % p
        ny = (cc * (self.x - nx) - self.y) % p
        return c.point(nx, ny)

    def dbl(self):
        """short.js:394-412 affine tangent (1 inversion)."""
        c = self.curve
        p = c.p
        if self.inf:
            return self
        ys1 = (2 * self.y) % p
        if ys1 == 0:
            return c.point(None, None)
        cc = (3 * self.x * self.x + c.a) * pow(ys1, -1, p) % p
        nx = (cc * cc - 2 * self.x) % p
        ny = (cc * (self.x - nx) - self.y) % p
        return c.point(nx, ny)

    def to_j(self) -> Jac:
        return (1, 1, 0) if self.inf else (self.x, self.y, 1)

    # --- tables: base.js:312-370 ---------------------------------------
    def get_naf_points(self, wnd):
        """base.js:357-370 (odd multiples by AFFINE adds, one inversion each)."""
        if self.pre is not None:
            return self.pre.naf_wnd, self.pre.naf_points
        res = [self]
        mx = (1 << wnd) - 1
        d = None if mx == 1 else self.dbl()
        for i in range(1, mx):
            res.append(res[i - 1].add(d))
        return wnd, res

    def get_doubles(self, step, power):
        """base.js:341-355."""
        if self.pre is not None:
            return self.pre.doubles_step, self.pre.doubles_points
        doubles = [self]
        acc = self
        for _ in range(0, power, step):
            for _ in range(step):
                acc = acc.dbl()
            doubles.append(acc)
        return step, doubles

    def precompute(self, power, wnd=8):
        """base.js:312-327 (wnd = 8 there; the table secp256k1 SHIPS has wnd 7,
        precomputed/secp256k1.js:268 -- on the curve the window does not change a
        result, off the curve, where the reference still computes, it does)."""
        if self.pre is not None:
            return self
        w, pts = self.get_naf_points(wnd)
        s, dbl = self.get_doubles(4, power)
        self.pre = _Tables(w, pts, s, dbl)
        return self

    def has_doubles(self, k: int) -> bool:
        """base.js:329-338."""
        if self.pre is None:
            return False
        return len(self.pre.doubles_points) >= -(-(k.bit_length() + 1) // self.pre.doubles_step)

    def get_beta(self):
        """short.js:282-310: (beta*x, y), tables mapped likewise."""
        c = self.curve
        if c.endo is None:
            return None
        if self._beta is not None:
            return self._beta
        b = ShortPoint(c, self.x * c.endo["beta"] % c.p, self.y)
        if self.pre is not None:
            mp = lambda q: ShortPoint(c, q.x * c.endo["beta"] % c.p, q.y)
            b.pre = _Tables(self.pre.naf_wnd, [mp(q) for q in self.pre.naf_points],
                            self.pre.doubles_step,
                            [mp(q) for q in self.pre.doubles_points])
        self._beta = b
        return b

    # --- public hot-path entry points ------------------------------------
    def mul(self, k: int):
        """short.js:422-432."""
        c = self.curve
        if self.inf:
            return self
        if self.has_doubles(k):
            return c.fixed_naf_mul(self, k)
        if c.endo is not None:
            return c.endo_wnaf_mul_add([self], [k], False)
        return c.wnaf_mul(self, k)

    def mul_add(self, k1: int, p2, k2: int):
        """short.js:434-441."""
        c = self.curve
        if c.endo is not None:
            return c.endo_wnaf_mul_add([self, p2], [k1, k2], False)
        return c.wnaf_mul_add(1, [self, p2], [k1, k2], 2, False)

    def jmul_add(self, k1: int, p2, k2: int) -> Jac:
        """short.js:443-450 (Jacobian result)."""
        c = self.curve
        if c.endo is not None:
            return c.endo_wnaf_mul_add([self, p2], [k1, k2], True)
        return c.wnaf_mul_add(1, [self, p2], [k1, k2], 2, True)


class ShortCurve:
    """lib/elliptic/curve/short.js ShortCurve + base.js BaseCurve."""

    def __init__(self, name, p, a, b, n, gx, gy, endo=None):
        self.name = name
        self.type = "short"
        self.p, self.a, self.b, self.n = p, a % p, b % p, n
        self.zero_a = self.a == 0                       # short.js:17
        self.three_a = (self.a + 3) % p == 0            # short.js:18
        self.bit_length = n.bit_length()                # base.js:31
        self.bytes = (p.bit_length() + 7) // 8
        self.endo = endo
        # base.js:33-40 (Maxwell trick enabled when p/n <= 100)
        self.maxwell = (p // n) <= 100
        self.g = ShortPoint(self, gx, gy)

    def point(self, x, y):
        return ShortPoint(self, x, y)

    def point_from_x(self, x: int, odd: bool) -> ShortPoint:
        """short.js:187-204 pointFromX: y = sqrt(x^3 + a x + b), parity fixed up;
        raises ValueError('invalid point') when x is not on the curve."""
        p = self.p
        x %= p
        y2 = (x * x % p * x + self.a * x + self.b) % p
        y = _sqrt_mod(y2, p)
        if y is None or (y * y - y2) % p != 0:
            raise ValueError("invalid point")
        if bool(y & 1) != bool(odd):
            y = (-y) % p
        return self.point(x, y)

    def validate(self, pt: ShortPoint) -> bool:
        """short.js:206-216."""
        if pt.inf:
            return True
        p = self.p
        return (pt.y * pt.y - (pt.x ** 3 + self.a * pt.x + self.b)) % p == 0

    # --- Jacobian formulas: short.js:516-925 -----------------------------
    def j_is_inf(self, P: Jac) -> bool:
        return P[2] % self.p == 0

    def j_to_p(self, P: Jac) -> ShortPoint:
        """short.js:516-526."""
        if self.j_is_inf(P):
            return self.point(None, None)
        p = self.p
        zinv = pow(P[2], -1, p)
        zinv2 = zinv * zinv % p
        return self.point(P[0] * zinv2 % p, P[1] * zinv2 * zinv % p)

    def j_add(self, P: Jac, Q: Jac) -> Jac:
        """short.js:532-567 (12M+4S; h==0 -> infinity or doubling)."""
        p = self.p
        if self.j_is_inf(P):
            return Q
        if self.j_is_inf(Q):
            return P
        pz2 = Q[2] * Q[2] % p
        z2 = P[2] * P[2] % p
        u1 = P[0] * pz2 % p
        u2 = Q[0] * z2 % p
        s1 = P[1] * pz2 * Q[2] % p
        s2 = Q[1] * z2 * P[2] % p
        h = (u1 - u2) % p
        r = (s1 - s2) % p
        if h == 0:
            if r != 0:
                return (1, 1, 0)
            return self.j_dbl(P)
        h2 = h * h % p
        h3 = h2 * h % p
        v = u1 * h2 % p
        nx = (r * r + h3 - 2 * v) % p
        ny = (r * (v - nx) - s1 * h3) % p
        nz = P[2] * Q[2] * h % p
        return (nx, ny, nz)

    def j_mixed_add(self, P: Jac, q: ShortPoint) -> Jac:
        """short.js:569-603 (8M+3S)."""
        p = self.p
        if self.j_is_inf(P):
            return q.to_j()
        if q.inf:
            return P
        z2 = P[2] * P[2] % p
        u1 = P[0]
        u2 = q.x * z2 % p
        s1 = P[1]
        s2 = q.y * z2 * P[2] % p
        h = (u1 - u2) % p
        r = (s1 - s2) % p
        if h == 0:
            if r != 0:
                return (1, 1, 0)
            return self.j_dbl(P)
        h2 = h * h % p
        h3 = h2 * h % p
        v = u1 * h2 % p
        nx = (r * r + h3 - 2 * v) % p
        ny = (r * (v - nx) - s1 * h3) % p
        nz = P[2] * h % p
        return (nx, ny, nz)

    def j_dbl(self, P: Jac) -> Jac:
        """short.js:656-830: _zeroDbl (a=0) / _threeDbl (a=-3) / _dbl.

        The three reference variants are algebraically the same doubling
        (dbl-2009-l / dbl-2001-b / dbl-2007-bl of the EFD); their (X,Y,Z) are
        projectively equal, and only the affine image is canonical (SURVEY
        8b), so one formula per case with the reference's Z3 = 2*Y1*Z1.
        """
        p = self.p
        if self.j_is_inf(P):
            return P
        X, Y, Z = P
        yy = Y * Y % p
        s = 4 * X * yy % p
        if self.zero_a:
            m = 3 * X * X % p
        elif self.three_a:
            zz = Z * Z % p
            m = 3 * (X - zz) * (X + zz) % p
        else:
            zz = Z * Z % p
            m = (3 * X * X + self.a * zz * zz) % p
        nx = (m * m - 2 * s) % p
        ny = (m * (s - nx) - 8 * yy * yy) % p
        nz = 2 * Y * Z % p
        return (nx, ny, nz)

    def j_dblp(self, P: Jac, pw: int) -> Jac:
        """short.js:605-654 (repeated doubling)."""
        for _ in range(pw):
            P = self.j_dbl(P)
        return P

    def j_eq_x_to_p(self, P: Jac, x: int) -> bool:
        """short.js:908-925 eqXToP: X == x*Z^2, retrying x += n while x < p."""
        p = self.p
        zs = P[2] * P[2] % p
        rx = x % p * zs % p
        if P[0] % p == rx:
            return True
        xc = x
        t = self.n % p * zs % p
        while True:
            xc += self.n
            if xc >= p:
                return False
            rx = (rx + t) % p
            if P[0] % p == rx:
                return True

    # --- ladders: base.js ----------------------------------------------------
    def fixed_naf_mul(self, pt: ShortPoint, k: int) -> ShortPoint:
        """base.js:52-84 _fixedNafMul (comb over 'doubles', step 4)."""
        step, pts = pt.get_doubles(0, 0)
        naf = get_naf(k, 1, self.bit_length)
        I = ((1 << (step + 1)) - (2 if step % 2 == 0 else 1)) // 3
        rep = []
        for j in range(0, len(naf), step):
            w = 0
            for l in range(j + step - 1, j - 1, -1):
                w = (w << 1) + (naf[l] if l < len(naf) else 0)
            rep.append(w)
        a: Jac = (1, 1, 0)
        b: Jac = (1, 1, 0)
        for i in range(I, 0, -1):
            for j, w in enumerate(rep):
                if w == i:
                    b = self.j_mixed_add(b, pts[j])
                elif w == -i:
                    b = self.j_mixed_add(b, pts[j].neg())
            a = self.j_add(a, b)
        return self.j_to_p(a)

    def wnaf_mul(self, pt: ShortPoint, k: int) -> ShortPoint:
        """base.js:86-126 _wnafMul (w=4 unless the point carries tables)."""
        w, wnd = pt.get_naf_points(4)
        naf = get_naf(k, w, self.bit_length)
        acc: Jac = (1, 1, 0)
        i = len(naf) - 1
        while i >= 0:
            l = 0
            while i >= 0 and naf[i] == 0:
                l += 1
                i -= 1
            if i >= 0:
                l += 1
            acc = self.j_dblp(acc, l)
            if i < 0:
                break
            z = naf[i]
            q = wnd[(z - 1) >> 1] if z > 0 else wnd[(-z - 1) >> 1].neg()
            acc = self.j_mixed_add(acc, q)
            i -= 1
        return self.j_to_p(acc)

    def wnaf_mul_add(self, def_w, points, coeffs, ln, jacobian_result):
        """base.js:128-253 _wnafMulAdd (interleaved wNAF, JSF for w=1 pairs)."""
        wnd_w = [0] * ln
        wnd: list = [None] * ln
        naf: list = [None] * ln
        mx = 0
        for i in range(ln):
            wnd_w[i], wnd[i] = points[i].get_naf_points(def_w)
        i = ln - 1
        while i >= 1:
            a, b = i - 1, i
            if wnd_w[a] != 1 or wnd_w[b] != 1:
                naf[a] = get_naf(coeffs[a], wnd_w[a], self.bit_length)
                naf[b] = get_naf(coeffs[b], wnd_w[b], self.bit_length)
                mx = max(len(naf[a]), len(naf[b]), mx)
                i -= 2
                continue
            comb = [points[a], None, None, points[b]]
            if points[a].y == points[b].y:                       # base.js:174
                comb[1] = points[a].add(points[b])
                comb[2] = ("j", self.j_mixed_add(points[a].to_j(), points[b].neg()))
            elif points[a].y == (-points[b].y) % self.p:
                comb[1] = ("j", self.j_mixed_add(points[a].to_j(), points[b]))
                comb[2] = points[a].add(points[b].neg())
            else:
                comb[1] = ("j", self.j_mixed_add(points[a].to_j(), points[b]))
                comb[2] = ("j", self.j_mixed_add(points[a].to_j(), points[b].neg()))
            index = [-3, -1, -5, -7, 0, 7, 5, 1, 3]
            jsf = get_jsf(coeffs[a], coeffs[b])
            mx = max(len(jsf[0]), mx)
            naf[a] = [0] * mx
            naf[b] = [0] * mx
            for j in range(mx):
                ja = jsf[0][j] if j < len(jsf[0]) else 0
                jb = jsf[1][j] if j < len(jsf[1]) else 0
                naf[a][j] = index[(ja + 1) * 3 + (jb + 1)]
            wnd[a] = comb
            i -= 2
        if ln % 2 == 1 and naf[0] is None:
            # odd count: base.js leaves naf[0] unset only when ln == 1 is never
            # called that way by the reference; guard for completeness.
            naf[0] = get_naf(coeffs[0], wnd_w[0], self.bit_length)
            mx = max(len(naf[0]), mx)

        def digit(j, i):
            return naf[j][i] if i < len(naf[j]) else 0

        acc: Jac = (1, 1, 0)
        i = mx
        while i >= 0:
            k = 0
            tmp = [0] * ln
            while i >= 0:
                zero = True
                for j in range(ln):
                    tmp[j] = digit(j, i)
                    if tmp[j] != 0:
                        zero = False
                if not zero:
                    break
                k += 1
                i -= 1
            if i >= 0:
                k += 1
            acc = self.j_dblp(acc, k)
            if i < 0:
                break
            for j in range(ln):
                z = tmp[j]
                if z == 0:
                    continue
                q = wnd[j][(z - 1) >> 1] if z > 0 else wnd[j][(-z - 1) >> 1]
                neg = z < 0
                if isinstance(q, tuple):                   # Jacobian comb entry
                    jq = q[1]
                    if neg:
                        jq = (jq[0], (-jq[1]) % self.p, jq[2])
                    acc = self.j_add(acc, jq)
                else:
                    acc = self.j_mixed_add(acc, q.neg() if neg else q)
            i -= 1
        return acc if jacobian_result else self.j_to_p(acc)

    def endo_split(self, k: int):
        """short.js:168-185 _endoSplit."""
        v1, v2 = self.endo["basis"]
        c1 = div_round(v2[1] * k, self.n)
        c2 = div_round(-v1[1] * k, self.n)
        k1 = k - c1 * v1[0] - c2 * v2[0]
        k2 = -(c1 * v1[1] + c2 * v2[1])
        return k1, k2

    def endo_wnaf_mul_add(self, points, coeffs, jacobian_result):
        """short.js:218-249 _endoWnafMulAdd (GLV)."""
        npoints, ncoeffs = [], []
        for pt, k in zip(points, coeffs):
            k1, k2 = self.endo_split(k)
            beta = pt.get_beta()
            if k1 < 0:
                k1 = -k1
                pt = pt.neg(True)
            if k2 < 0:
                k2 = -k2
                beta = beta.neg(True)
            npoints += [pt, beta]
            ncoeffs += [k1, k2]
        return self.wnaf_mul_add(1, npoints, ncoeffs, len(npoints), jacobian_result)


# --------------------------------------------------------------------------
# ECDSA verify -- lib/elliptic/ec/index.js
# --------------------------------------------------------------------------

def truncate_to_n(curve: ShortCurve, msg: int, byte_length: int, trunc_only=False,
                  bit_length=None) -> int:
    """ec/index.js:81-108 _truncateToN.  `byte_length` is the length of the
    caller's buffer / hex string (what 6.6.1 uses, :84-96); `bit_length`
    mirrors options.msgBitLength."""
    if bit_length is None:
        bit_length = byte_length * 8
    delta = bit_length - curve.n.bit_length()
    if delta > 0:
        msg >>= delta
    if not trunc_only and msg >= curve.n:
        return msg - curve.n
    return msg


def ecdsa_verify(curve: ShortCurve, msg: int, msg_bytes: int, r: int, s: int,
                 pub: ShortPoint, msg_bit_length=None) -> bool:
    """ec/index.js:188-229 EC#verify (after key/signature decoding).

    `curve.g` must carry its tables (EC's constructor calls g.precompute,
    ec/index.js:36) for the ladder to take the reference's path; the result
    does not depend on it.
    """
    n = curve.n
    msg = truncate_to_n(curve, msg, msg_bytes, False, msg_bit_length)
    if r < 1 or r >= n:
        return False
    if s < 1 or s >= n:
        return False
    sinv = pow(s, -1, n)
    u1 = sinv * msg % n
    u2 = sinv * r % n
    if not curve.maxwell:
        pt = curve.g.mul_add(u1, pub, u2)
        if pt.inf:
            return False
        return pt.x % n == r
    jp = curve.g.jmul_add(u1, pub, u2)
    if curve.j_is_inf(jp):
        return False
    return curve.j_eq_x_to_p(jp, r)


CURVE_HASH = {"secp256k1": "sha256", "p192": "sha256", "p224": "sha256", "p256": "sha256",
              "p384": "sha384", "p521": "sha512"}        # lib/elliptic/curves.js: `hash:` of each preset


class HmacDrbg:
    """hmac-drbg 1.0.1 lib/hmac-drbg.js (the reference's dependency, pinned in
    package-lock.json): HMAC_DRBG of NIST SP 800-90A without prediction resistance, as
    EC#sign instantiates it (entropy = private key, nonce = truncated message, no
    personalisation string)."""

    def __init__(self, hash_name: str, entropy: bytes, nonce: bytes):
        import hashlib
        import hmac
        self._new = lambda key: hmac.new(key, digestmod=getattr(hashlib, hash_name))
        out = getattr(hashlib, hash_name)().digest_size
        self.K = b"\x00" * out                     # _init :37-48
        self.V = b"\x01" * out
        self._update(entropy + nonce)

    def _hmac(self, *parts) -> bytes:
        h = self._new(self.K)
        for p in parts:
            h.update(p)
        return h.digest()

    def _update(self, seed: bytes = b""):            # :54-69
        self.K = self._hmac(self.V, b"\x00", seed)
        self.V = self._hmac(self.V)
        if not seed:
            return
        self.K = self._hmac(self.V, b"\x01", seed)
        self.V = self._hmac(self.V)

    def generate(self, n: int) -> bytes:             # :91-113
        temp = b""
        while len(temp) < n:
            self.V = self._hmac(self.V)
            temp += self.V
        self._update()
        return temp[:n]


def ecdsa_sign_det(curve: ShortCurve, name: str, msg: int, msg_bytes: int, d: int, canonical=False):
    """ec/index.js:110-186 EC#sign with its own nonce source: the loop over
    drbg.generate(n.byteLength()) until a nonce is accepted.  -> (r, s, recoveryParam)"""
    nb = (curve.n.bit_length() + 7) // 8
    e = truncate_to_n(curve, msg, msg_bytes, False)
    drbg = HmacDrbg(CURVE_HASH[name], d.to_bytes(nb, "big"), e.to_bytes(nb, "big"))
    while True:
        res = ecdsa_sign(curve, msg, msg_bytes, d, drbg.generate(nb), canonical)
        if res is not None:
            return res


def ecdsa_recover(curve: ShortCurve, e: int, r: int, s: int, j: int) -> ShortPoint:
    """ec/index.js:231-259 EC#recoverPubKey: Q = r^-1 (s R - e G), R = the point with
    x = r (+ n when j & 2) and y-parity j & 1.  e = new BN(msg) is used as it is (no
    _truncateToN here), only reduced by the arithmetic mod n.  Raises ValueError with the
    reference's message where it throws."""
    if (3 & j) != j:
        raise ValueError("The recovery param is more than two bits")
    n = curve.n
    is_y_odd = j & 1
    is_second = j >> 1
    if r >= curve.p % n and is_second:
        raise ValueError("Unable to find sencond key candinate")
    R = curve.point_from_x(r + n if is_second else r, bool(is_y_odd))       # 'invalid point'
    rinv = pow(r, -1, n)
    s1 = (n - e) * rinv % n
    s2 = s * rinv % n
    return curve.g.mul_add(s1, R, s2)


def _der_get_length(buf: bytes, place: int):
    """signature.js:30-59 getLength -> (value or None for the reference's `false`, place).
    Reads past the end behave like JavaScript's `undefined` (| 0 -> 0, comparisons fail)."""
    if place >= len(buf):
        return "undefined", place + 1
    initial = buf[place]
    place += 1
    if not initial & 0x80:
        return initial, place
    octets = initial & 0xF
    if octets == 0 or octets > 4:
        return None, place
    if place < len(buf) and buf[place] == 0:
        return None, place
    val = 0
    off = place
    for _ in range(octets):
        val = ((val << 8) | (buf[off] if off < len(buf) else 0)) & 0xFFFFFFFF
        off += 1
    if val <= 0x7F:
        return None, place
    return val, off


def sig_import_der(data: bytes) -> Optional[Tuple[int, int]]:
    """signature.js:83-147 Signature#_importDER -> (r, s), or None where it returns false (the
    constructor then throws 'Signature without r or s')."""
    d = bytes(data)

    def at(i):
        return d[i] if 0 <= i < len(d) else None
    p = 0
    if at(p) != 0x30:
        return None
    p += 1
    ln, p = _der_get_length(d, p)
    if ln is None or ln == "undefined" or ln + p != len(d):
        return None
    if at(p) != 0x02:
        return None
    p += 1
    rlen, p = _der_get_length(d, p)
    if rlen is None or rlen == "undefined":
        return None                       # `undefined` runs into the next tag test with place = NaN
    if (at(p) or 0) & 0x80:
        return None
    r = d[p:p + rlen]
    p += rlen
    if at(p) != 0x02:
        return None
    p += 1
    slen, p = _der_get_length(d, p)
    if slen is None or slen == "undefined" or len(d) != slen + p:
        return None
    if (at(p) or 0) & 0x80:
        return None
    s = d[p:p + slen]
    if len(r) and r[0] == 0:
        if len(r) > 1 and r[1] & 0x80:
            r = r[1:]
        else:
            return None
    if len(s) and s[0] == 0:
        if len(s) > 1 and s[1] & 0x80:
            s = s[1:]
        else:
            return None
    return int.from_bytes(r, "big"), int.from_bytes(s, "big")


def sig_to_der(r: int, s: int) -> bytes:
    """signature.js:149-176 Signature#toDER (s = 0 does not terminate in the reference)"""
    assert s != 0

    def integer(v):
        b = v.to_bytes(max(1, (v.bit_length() + 7) // 8), "big")
        if b[0] & 0x80:
            b = b"\x00" + b
        return b

    def length(n):
        if n < 0x80:
            return bytes([n])
        octets = 1 + ((n.bit_length() - 1) >> 3)
        return bytes([0x80 | octets]) + n.to_bytes(octets, "big")
    rb, sb = integer(r), integer(s)
    body = b"\x02" + length(len(rb)) + rb + b"\x02" + length(len(sb)) + sb
    return b"\x30" + length(len(body)) + body


def ecdsa_verify_wire(curve: ShortCurve, msg: int, msg_bytes: int, der: bytes, key: bytes) -> bool:
    """ec/index.js:188-229 EC#verify(msg, der, key): keyFromPublic -> decodePoint, then
    new Signature(der); raises ValueError with the reference's message where it throws."""
    pub = decode_point(curve, key)
    rs = sig_import_der(der)
    if rs is None:
        raise ValueError("Signature without r or s")
    return ecdsa_verify(curve, msg, msg_bytes, rs[0], rs[1], pub)


def decode_point(curve: ShortCurve, data: bytes) -> ShortPoint:
    """base.js:270-293 BaseCurve#decodePoint.  Raises ValueError with the reference's message:
    'Unknown point format', 'Assertion failed' (hybrid prefix vs. y's last bit), 'invalid point'
    (pointFromX).  Uncompressed coordinates are reduced mod p (short.js:261-264) and NOT checked
    against the curve equation."""
    ln = (curve.p.bit_length() + 7) // 8
    b = bytes(data)
    if len(b) and b[0] in (4, 6, 7) and len(b) - 1 == 2 * ln:
        if b[0] == 6 and b[-1] % 2 != 0:
            raise ValueError("Assertion failed")
        if b[0] == 7 and b[-1] % 2 != 1:
            raise ValueError("Assertion failed")
        return curve.point(int.from_bytes(b[1:1 + ln], "big") % curve.p,
                           int.from_bytes(b[1 + ln:], "big") % curve.p)
    if len(b) and b[0] in (2, 3) and len(b) - 1 == ln:
        return curve.point_from_x(int.from_bytes(b[1:], "big"), b[0] == 3)
    raise ValueError("Unknown point format")


def encode_point(curve: ShortCurve, pt: ShortPoint, compact: bool) -> bytes:
    """base.js:299-311 BasePoint#_encode"""
    ln = (curve.p.bit_length() + 7) // 8
    x = (pt.x % curve.p).to_bytes(ln, "big")
    if compact:
        return bytes([3 if (pt.y % curve.p) & 1 else 2]) + x
    return b"\x04" + x + (pt.y % curve.p).to_bytes(ln, "big")


def key_validate(curve: ShortCurve, pt: ShortPoint) -> Tuple[bool, Optional[str]]:
    """ec/key.js:41-52 KeyPair#validate -> (result, reason)"""
    if pt.inf:
        return False, "Invalid public key"
    if not curve.validate(pt):
        return False, "Public key is not a point"
    if not pt.mul(curve.n).inf:
        return False, "Public key * N != O"
    return True, None


def ecdsa_sign(curve: ShortCurve, msg: int, msg_bytes: int, d: int, k_bytes: bytes, canonical=False,
               msg_bit_length=None):
    """ec/index.js:110-186 EC#sign for ONE supplied nonce (options.k(0)): returns
    (r, s, recoveryParam), or None where the reference would move on to the next nonce."""
    n = curve.n
    e = truncate_to_n(curve, msg, msg_bytes, False, msg_bit_length)
    kv = int.from_bytes(k_bytes, "big")
    # :153-156: the nonce reaches _truncateToN as a BN, so its byte length is that of the VALUE
    k = truncate_to_n(curve, kv, (kv.bit_length() + 7) // 8, True)
    if k <= 1 or k >= n - 1:                                                         # :157-158
        return None
    kp = curve.g.mul(k)
    if kp.inf:
        return None
    r = kp.x % n
    if r == 0:
        return None
    s = pow(k, -1, n) * (r * d + e) % n
    if s == 0:
        return None
    recid = (1 if kp.y & 1 else 0) | (2 if kp.x != r else 0)
    if canonical and s > n >> 1:
        s = n - s
        recid ^= 1
    return r, s, recid


# --------------------------------------------------------------------------
# Twisted Edwards -- lib/elliptic/curve/edwards.js  (extended coords, a = -1)
# --------------------------------------------------------------------------

class EdPoint:
    """edwards.js:114-153 Point: extended (X, Y, Z, T)."""

    __slots__ = ("curve", "x", "y", "z", "t", "pre")

    def __init__(self, curve, x, y, z=None, t=None):
        p = curve.p
        self.curve = curve
        if x is None and y is None:
            self.x, self.y, self.z, self.t = 0, 1, 1, 0
        else:
            self.x, self.y = x % p, y % p
            self.z = 1 if z is None else z % p
            if t is None:
                # edwards.js:141-147: T = X*Y / Z
                self.t = self.x * self.y % p
                if self.z != 1:
                    self.t = self.t * pow(self.z, -1, p) % p
            else:
                self.t = t % p
        self.pre: Optional[_Tables] = None

    def is_infinity(self):
        """edwards.js:167-172."""
        return self.x == 0 and (self.y == self.z)

    def neg(self):
        c = self.curve
        return EdPoint(c, (-self.x) % c.p, self.y, self.z, (-self.t) % c.p)

    def dbl(self):
        """edwards.js:174-205 _extDbl (dbl-2008-hwcd, 4M+4S)."""
        if self.is_infinity():
            return self
        c = self.curve
        p = c.p
        a = self.x * self.x % p
        b = self.y * self.y % p
        cc = 2 * self.z * self.z % p
        d = c.a * a % p
        e = ((self.x + self.y) ** 2 - a - b) % p
        g = (d + b) % p
        f = (g - cc) % p
        h = (d - b) % p
        return EdPoint(c, e * f % p, g * h % p, f * g % p, e * h % p)

    def add(self, o):
        """edwards.js:350-360 add -> :279-309 _extAdd (add-2008-hwcd-3, 8M)."""
        if self.is_infinity():
            return o
        if o.is_infinity():
            return self
        c = self.curve
        p = c.p
        a = (self.y - self.x) * (o.y - o.x) % p
        b = (self.y + self.x) * (o.y + o.x) % p
        cc = self.t * c.dd % p * o.t % p
        d = self.z * 2 * o.z % p
        e = (b - a) % p
        f = (d - cc) % p
        g = (d + cc) % p
        h = (b + a) % p
        return EdPoint(c, e * f % p, g * h % p, f * g % p, e * h % p)

    def normalized(self) -> Tuple[int, int]:
        """edwards.js:377-390 normalize + getX/getY."""
        p = self.curve.p
        zi = pow(self.z, -1, p)
        return self.x * zi % p, self.y * zi % p

    def get_naf_points(self, wnd):
        if self.pre is not None:
            return self.pre.naf_wnd, self.pre.naf_points
        res = [self]
        mx = (1 << wnd) - 1
        d = None if mx == 1 else self.dbl()
        for i in range(1, mx):
            res.append(res[i - 1].add(d))
        return wnd, res

    def precompute(self, power):
        if self.pre is not None:
            return self
        w, pts = self.get_naf_points(8)
        doubles = [self]
        acc = self
        for _ in range(0, power, 4):
            for _ in range(4):
                acc = acc.dbl()
            doubles.append(acc)
        self.pre = _Tables(w, pts, 4, doubles)
        return self

    def has_doubles(self, k):
        if self.pre is None:
            return False
        return len(self.pre.doubles_points) >= -(-(k.bit_length() + 1) // 4)

    def mul(self, k: int):
        """edwards.js:362-367."""
        c = self.curve
        if self.has_doubles(k):
            return c.fixed_naf_mul(self, k)
        return c.wnaf_mul(self, k)

    def mul_add(self, k1, o, k2):
        """edwards.js:369-371 (reference only works when the pair is not
        (wnd 1, wnd 1): base.js:175 calls toJ(), absent on Edwards points)."""
        return self.curve.wnaf_mul_add(1, [self, o], [k1, k2], 2)


class EdwardsCurve:
    """lib/elliptic/curve/edwards.js EdwardsCurve (twisted, a=-1, c=1)."""

    def __init__(self, name, p, a, d, n, gx, gy):
        self.name = name
        self.type = "edwards"
        self.p, self.a, self.d, self.n = p, a % p, d % p, n
        self.dd = 2 * self.d % p                        # edwards.js:23
        self.bit_length = n.bit_length()
        self.bytes = (p.bit_length() + 7) // 8
        self.g = EdPoint(self, gx, gy)

    def point(self, x, y, z=None, t=None):
        return EdPoint(self, x, y, z, t)

    def point_from_y(self, y: int, odd: bool) -> EdPoint:
        """edwards.js:71-97 pointFromY (c = 1)."""
        p = self.p
        y %= p
        y2 = y * y % p
        lhs = (y2 - 1) % p
        rhs = (y2 * self.d - self.a) % p                # c2 = 1
        if rhs == 0:
            raise ValueError("invalid point")            # reference: redInvm(0) asserts
        x2 = lhs * pow(rhs, -1, p) % p
        if x2 == 0:
            if odd:
                raise ValueError("invalid point")
            return self.point(0, y)
        x = _sqrt_mod(x2, p)
        if x is None:
            raise ValueError("invalid point")
        if (x & 1) != int(odd):
            x = (-x) % p
        return self.point(x, y)

    def fixed_naf_mul(self, pt: EdPoint, k: int) -> EdPoint:
        """base.js:52-84 with mixedAdd == add == _extAdd (edwards.js:434-435)."""
        step, pts = pt.pre.doubles_step, pt.pre.doubles_points
        naf = get_naf(k, 1, self.bit_length)
        I = ((1 << (step + 1)) - (2 if step % 2 == 0 else 1)) // 3
        rep = []
        for j in range(0, len(naf), step):
            w = 0
            for l in range(j + step - 1, j - 1, -1):
                w = (w << 1) + (naf[l] if l < len(naf) else 0)
            rep.append(w)
        a = self.point(None, None)
        b = self.point(None, None)
        for i in range(I, 0, -1):
            for j, w in enumerate(rep):
                if w == i:
                    b = b.add(pts[j])
                elif w == -i:
                    b = b.add(pts[j].neg())
            a = a.add(b)
        return a

    def wnaf_mul(self, pt: EdPoint, k: int) -> EdPoint:
        """base.js:86-126 for type 'projective' (returns un-normalized)."""
        w, wnd = pt.get_naf_points(4)
        naf = get_naf(k, w, self.bit_length)
        acc = self.point(None, None)
        i = len(naf) - 1
        while i >= 0:
            l = 0
            while i >= 0 and naf[i] == 0:
                l += 1
                i -= 1
            if i >= 0:
                l += 1
            for _ in range(l):
                acc = acc.dbl()
            if i < 0:
                break
            z = naf[i]
            q = wnd[(z - 1) >> 1] if z > 0 else wnd[(-z - 1) >> 1].neg()
            acc = acc.add(q)
            i -= 1
        return acc

    def wnaf_mul_add(self, def_w, points, coeffs, ln):
        """base.js:128-253 restricted to the branch Edwards can take
        (no (1,1) JSF pair, see EdPoint.mul_add)."""
        wnd_w, wnd, naf = [0] * ln, [None] * ln, [None] * ln
        for i in range(ln):
            wnd_w[i], wnd[i] = points[i].get_naf_points(def_w)
        if ln == 2 and wnd_w[0] == 1 and wnd_w[1] == 1:
            raise TypeError("points[a].toJ is not a function")   # reference behaviour
        mx = 0
        for i in range(ln):
            naf[i] = get_naf(coeffs[i], wnd_w[i], self.bit_length)
            mx = max(mx, len(naf[i]))
        acc = self.point(None, None)
        for i in range(mx, -1, -1):
            acc = acc.dbl()
            for j in range(ln):
                z = naf[j][i] if i < len(naf[j]) else 0
                if z == 0:
                    continue
                q = wnd[j][(z - 1) >> 1] if z > 0 else wnd[j][(-z - 1) >> 1].neg()
                acc = acc.add(q)
        return acc


def ed_decode_point(curve: "EdwardsCurve", data: bytes) -> EdPoint:
    """eddsa/index.js:99-109 decodePoint: y = LE int with the top bit cleared, top bit = x parity"""
    b = bytearray(data)
    odd = (b[-1] & 0x80) != 0
    b[-1] &= 0x7F
    return curve.point_from_y(int.from_bytes(bytes(b), "little"), odd)


def ed_validate(curve: "EdwardsCurve", x: int, y: int) -> bool:
    """edwards.js:99-112 EdwardsCurve#validate of an affine point (c = 1)"""
    p = curve.p
    x2, y2 = x * x % p, y * y % p
    return (x2 * curve.a + y2 - (1 + curve.d * x2 % p * y2)) % p == 0


def eddsa_verify(curve: "EdwardsCurve", msg: bytes, sig: bytes, pub: bytes) -> bool:
    """eddsa/index.js:52-63 EDDSA#verify (ed25519, SHA-512).  Raises ValueError where the
    reference throws (an R or A that does not decode to a curve point)."""
    import hashlib
    assert len(sig) == 64 and len(pub) == 32
    S = int.from_bytes(sig[32:], "little")
    if S >= curve.n:
        return False
    h = int.from_bytes(hashlib.sha512(sig[:32] + pub + msg).digest(), "little") % curve.n   # hashInt :65-70
    SG = curve.g.mul(S)
    A = ed_decode_point(curve, pub)
    R = ed_decode_point(curve, sig[:32])
    lhs = R.add(A.mul(h))
    return lhs.normalized() == SG.normalized()


def ed_encode_point(P: "EdPoint") -> bytes:
    """eddsa/index.js:94-98 EDDSA#encodePoint: y little-endian, top bit = parity of x"""
    x, y = P.normalized()
    enc = bytearray(y.to_bytes(32, "little"))
    enc[31] |= 0x80 if x & 1 else 0
    return bytes(enc)


def eddsa_keypair(curve: "EdwardsCurve", secret: bytes) -> Tuple[int, bytes, bytes]:
    """eddsa/key.js:42-75 KeyPair from a secret: hash = SHA-512(secret); the first 32 bytes,
    clamped (:51-62), are the private scalar a (little-endian, :65-67); the rest is the message
    prefix (:73-75); the public key is encodePoint(G * a) (:42-49).  -> (a, prefix, pub)"""
    import hashlib
    h = hashlib.sha512(secret).digest()
    a = bytearray(h[:32])
    a[0] &= 248
    a[31] &= 127
    a[31] |= 64
    a_int = int.from_bytes(bytes(a), "little")
    return a_int, h[32:], ed_encode_point(curve.g.mul(a_int))


def eddsa_sign(curve: "EdwardsCurve", msg: bytes, secret: bytes) -> Tuple[bytes, bytes]:
    """eddsa/index.js:32-50 EDDSA#sign: r = H(prefix || M) mod n, R = G * r,
    S = (r + H(R || A || M) * a) mod n.  -> (signature R || S, public key A)"""
    import hashlib
    a, prefix, pub = eddsa_keypair(curve, secret)
    r = int.from_bytes(hashlib.sha512(prefix + msg).digest(), "little") % curve.n
    Renc = ed_encode_point(curve.g.mul(r))
    h = int.from_bytes(hashlib.sha512(Renc + pub + msg).digest(), "little") % curve.n
    S = (r + h * a) % curve.n
    return Renc + S.to_bytes(32, "little"), pub


def _sqrt_mod(a: int, p: int) -> Optional[int]:
    """bn.js Red.sqrt (dist/elliptic.js:7180-7230): p%4==3 -> pow; else
    Tonelli-Shanks.  Returns the root bn.js returns (either root is accepted by
    callers, which fix the parity afterwards)."""
    a %= p
    if a == 0:
        return 0
    if p % 4 == 3:
        r = pow(a, (p + 1) // 4, p)
        return r if r * r % p == a else None
    if pow(a, (p - 1) // 2, p) != 1:
        # a non-residue never reaches t^(2^i) == 1 with i < m in bn.js's Tonelli-Shanks loop:
        # `assert(i < m)` (dist/elliptic.js:7216) fails, the caller sees Error('Assertion failed')
        raise ValueError("Assertion failed")
    q, s = p - 1, 0
    while q % 2 == 0:
        q //= 2
        s += 1
    z = 2
    while pow(z, (p - 1) // 2, p) != p - 1:
        z += 1
    m, c, t, r = s, pow(z, q, p), pow(a, q, p), pow(a, (q + 1) // 2, p)
    while t != 1:
        i, t2 = 0, t
        while t2 != 1:
            t2 = t2 * t2 % p
            i += 1
        b = pow(c, 1 << (m - i - 1), p)
        m, c = i, b * b % p
        t, r = t * c % p, r * b % p
    return r


# --------------------------------------------------------------------------
# Montgomery x-only -- lib/elliptic/curve/mont.js
# --------------------------------------------------------------------------

class MontCurve:
    """lib/elliptic/curve/mont.js MontCurve (curve25519: A=486662, B=1)."""

    def __init__(self, name, p, a, b, gx, n=None):
        self.name = name
        self.type = "mont"
        self.p, self.a, self.b, self.n = p, a % p, b % p, n
        self.a24 = (self.a + 2) * pow(4, -1, p) % p      # mont.js:14-16
        self.bytes = (p.bit_length() + 7) // 8
        self.gx = gx

    def dbl(self, P):
        """mont.js:82-101 (dbl-1987-m-3)."""
        p = self.p
        X, Z = P
        aa = (X + Z) ** 2 % p
        bb = (X - Z) ** 2 % p
        c = (aa - bb) % p
        return aa * bb % p, c * (bb + self.a24 * c) % p

    def diff_add(self, P, Q, D):
        """mont.js:107-128 (dadd-1987-m-3): P+Q given D = P-Q."""
        p = self.p
        a, b = (P[0] + P[1]) % p, (P[0] - P[1]) % p
        c, d = (Q[0] + Q[1]) % p, (Q[0] - Q[1]) % p
        da, cb = d * a % p, c * b % p
        return D[1] * (da + cb) ** 2 % p, D[0] * (da - cb) ** 2 % p

    def mul(self, x: int, k: int):
        """mont.js:130-153 Point#mul: MSB-first over the exact bit length of
        k, no clamping.  Returns (X, Z)."""
        a = (x % self.p, 1)
        b = (1, 0)                                        # mont.js:32-34
        c = a
        for i in range(k.bit_length() - 1, -1, -1):
            if (k >> i) & 1 == 0:
                a = self.diff_add(a, b, c)
                b = self.dbl(b)
            else:
                b = self.diff_add(a, b, c)
                a = self.dbl(a)
        return b

    def mul_x(self, x: int, k: int) -> Optional[int]:
        """mul + getX (mont.js:167-178).  None when the result is infinity
        (Z == 0), where the reference's getX would throw on invm(0)."""
        X, Z = self.mul(x, k)
        if Z % self.p == 0:
            return None
        return X * pow(Z, -1, self.p) % self.p


# --------------------------------------------------------------------------
# Presets -- lib/elliptic/curves.js:43-206, loaded from the fixture that
# tools/gen_golden.js dumped out of the reference (tests/golden/curves.json)
# so that no constant here is hand-typed.
# --------------------------------------------------------------------------

_GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests",
                       "golden", "curves.json")
_CACHE: dict = {}


def _signed_hex(s: str) -> int:
    return -int(s[1:], 16) if s.startswith("-") else int(s, 16)


def get_curve(name: str, precompute: bool = True):
    """Instantiate a preset the way `new EC(name)` / `new EDDSA(name)` leaves
    it: G carries naf(8)+doubles(4) tables (ec/index.js:36, eddsa/index.js:19;
    secp256k1 ships them precomputed with naf(7), curves.js:169-205,
    precomputed/secp256k1.js:268)."""
    key = (name, precompute)
    if key in _CACHE:
        return _CACHE[key]
    with open(_GOLDEN) as f:
        c = json.load(f)[name]
    I = lambda s: int(s, 16)
    if c["type"] == "short":
        endo = None
        if "beta" in c:
            endo = {"beta": I(c["beta"]), "lambda": I(c["lambda"]),
                    "basis": [(_signed_hex(v["a"]), _signed_hex(v["b"])) for v in c["basis"]]}
        cur = ShortCurve(name, I(c["p"]), I(c["a"]), I(c["b"]), I(c["n"]), I(c["gx"]),
                         I(c["gy"]), endo)
        if precompute:
            cur.g.precompute(cur.n.bit_length() + 1, 7 if name == "secp256k1" else 8)
    elif c["type"] == "edwards":
        cur = EdwardsCurve(name, I(c["p"]), I(c["a"]), I(c["d"]), I(c["n"]), I(c["gx"]),
                           I(c["gy"]))
        if precompute:
            cur.g.precompute(cur.n.bit_length() + 1)
    else:
        cur = MontCurve(name, I(c["p"]), I(c["a"]), I(c["b"]), I(c["gx"]),
                        I(c["n"]) if "n" in c else None)
    _CACHE[key] = cur
    return cur


SHORT_CURVES = ["secp256k1", "p192", "p224", "p256", "p384", "p521"]
ALL_CURVES = SHORT_CURVES + ["ed25519", "curve25519"]
