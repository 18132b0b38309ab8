// ellgpu -- curve25519 x-only Montgomery ladder.
//
// Replaces lib/elliptic/curve/mont.js Point#mul (:130-153) + getX (:167-178):
//   dbl      <- Point#dbl     (mont.js:82-101,  dbl-1987-m-3,  2M+2S + a24)
//   diffadd  <- Point#diffAdd (mont.js:107-128, dadd-1987-m-3, 4M+2S)
// The reference walks the bits of k MSB-first over k's exact bit length, no
// clamping.  A lane cannot have its own trip count, so every lane walks all 256
// bits: while the leading bits are zero the pair (a, b) stays (~P, O) under the
// same formulas (O = (1, 0), mont.js:32-34), so the result is projectively
// identical.  The per-bit branch becomes two selects.
#pragma once

#include "curve_consts.h"
#include "fp.h"

namespace ell {

struct MontWork {
  typedef Fp25519 F;
  typedef F::El El;
  typedef consts::CURVE25519_C C;

  struct XZ {
    El x, z;
  };
  ELL_HD static XZ sel(bool c, const XZ& a, const XZ& b) {
    XZ r;
    bn_select<8>(r.x.v, c, a.x.v, b.x.v);
    bn_select<8>(r.z.v, c, a.z.v, b.z.v);
    return r;
  }
  // MontCurve#validate (mont.js:23-32): is x^3 + a x^2 + x a square?  The reference takes the root
  // (bn.js Red#sqrt: 0 for 0, Tonelli-Shanks otherwise -- which THROWS 'Assertion failed' on a
  // non-residue, dist/elliptic.js:7242-7302) and squares it back.  Here Euler's criterion:
  // rhs^((p-1)/2) = (rhs^(2^252-3))^4 rhs^2.  a = 4 a24 - 2.  F: Fp25519 or the row field Fp25519C.
  template <class FF>
  ELL_HD static bool x_has_point(const u32 (&xw)[8]) {
    typedef typename FF::El E;
    u32 aw[8] = {4u * C::a24[0] - 2u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const E x = FF::from_plain(xw);
    const E rhs = FF::mul(x, FF::add(FF::add(FF::sqr(x), FF::mul(x, FF::from_plain(aw))), FF::one()));
    const E w = FF::pow22523(rhs);
    const E e = FF::mul(FF::sqr(FF::sqr(w)), FF::sqr(rhs));
    u32 ew[8], rw[8];
    FF::to_plain(ew, e);
    FF::to_plain(rw, rhs);
    bool one = ew[0] == 1u;
    ELL_UNROLL
    for (int l = 1; l < 8; l++) one = one && ew[l] == 0u;
    return one || bn_is_zero<8>(rw);
  }
  ELL_HD static void validate(size_t i, const u8* xs, u8* out_bad) {
    u32 t[8];
    load_be<8>(t, xs + i * 32, 32);
    out_bad[i] = x_has_point<F>(t) ? 0 : 1;
  }
  // a24 = (a + 2) / 4 = 121666 is a one-limb constant on curve25519: a 1 x 8 limb product
  static_assert(C::a24[1] == 0 && C::a24[2] == 0 && C::a24[3] == 0 && C::a24[4] == 0 && C::a24[5] == 0 &&
                C::a24[6] == 0 && C::a24[7] == 0, "a24 must fit one limb");
  ELL_HD static XZ dbl(const XZ& p) {
    El aa = F::sqr(F::add(p.x, p.z));
    El bb = F::sqr(F::sub(p.x, p.z));
    El c = F::sub(aa, bb);
    XZ r;
    r.x = F::mul(aa, bb);
    r.z = F::mul(c, F::add(bb, F::mul_u32(c, C::a24[0])));
    return r;
  }
  // p + q given their difference (dx : 1)
  ELL_HD static XZ diffadd(const XZ& p, const XZ& q, const El& dx) {
    El a = F::add(p.x, p.z), b = F::sub(p.x, p.z);
    El c = F::add(q.x, q.z), d = F::sub(q.x, q.z);
    El da = F::mul(d, a), cb = F::mul(c, b);
    XZ r;
    r.x = F::sqr(F::add(da, cb));                 // * diff.z, which is 1
    r.z = F::mul(dx, F::sqr(F::sub(da, cb)));
    return r;
  }

  ELL_HD static void ladder(size_t i, size_t n, const u8* ks, const u8* xs, u32* xz) {
    u32 k[8], t[8];
    load_be<8>(k, ks + i * 32, 32);
    load_be<8>(t, xs + i * 32, 32);
    El x = F::from_plain(t);
    XZ a, b;
    a.x = x; a.z = F::one();                      // (N/2)*Q + Q
    b.x = F::one(); b.z = F::zero();              // (N/2)*Q
    ELL_NOUNROLL
    for (int w = 0; w < 256; w++) {
      bool bit = (k[7] >> 31) != 0;
      ELL_UNROLL
      for (int l = 7; l > 0; l--) k[l] = (k[l] << 1) | (k[l - 1] >> 31);
      k[0] <<= 1;
      XZ s = diffadd(a, b, x);
      XZ d = dbl(sel(bit, a, b));
      a = sel(bit, d, s);
      b = sel(bit, s, d);
    }
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {
      xz[(size_t)(0 * 8 + l) * n + i] = b.x.v[l];
      xz[(size_t)(1 * 8 + l) * n + i] = b.z.v[l];
    }
  }

  // getX: X / Z with one inversion per K items; Z == 0 -> out_inf = 1
  ELL_HD static void normalize(size_t t, size_t T, size_t n, int K, const u32* xz, u32* pre,
                               u8* out_x, u8* out_inf) {
    El acc = F::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      El z;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) z.v[l] = xz[(size_t)(8 + l) * n + i];
      bool inf = F::is_zero(z);
      bn_select<8>(z.v, inf, F::one().v, z.v);
      ELL_UNROLL
      for (int l = 0; l < 8; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = F::mul(acc, z);
    }
    El inv = F::inv(acc);
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      El X, Z, pr;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        X.v[l] = xz[(size_t)l * n + i];
        Z.v[l] = xz[(size_t)(8 + l) * n + i];
        pr.v[l] = pre[(size_t)l * n + i];
      }
      bool inf = F::is_zero(Z);
      bn_select<8>(Z.v, inf, F::one().v, Z.v);
      El zinv = F::mul(inv, pr);
      inv = F::mul(inv, Z);
      El x = F::mul(X, zinv);
      if (inf) x = F::zero();
      store_be<8>(out_x + i * 32, x.v, 32);
      out_inf[i] = inf ? 1 : 0;
    }
  }
};

}  // namespace ell
