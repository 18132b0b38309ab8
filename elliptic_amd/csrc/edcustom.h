// ellgpu -- user-defined (twisted) Edwards curves  a x^2 + y^2 = 1 + d x^2 y^2  over a run-time
// prime p < 2^256 (c = 1): `new elliptic.curve.edwards({p, a, c: 1, d, ...})` with parameters
// that are not ed25519's.  For a != -1 the reference works in projective coordinates
// (lib/elliptic/curve/edwards.js: _projDbl :207-266, _projAdd :311-348, normalize :377-390); the
// same formulas are used here for every a (the extended a = -1 forms of the reference compute the
// same group law, and only the affine image is canonical):
//   add   add-2008-bbjlp   10M + 1S + 1a + 1d     (unified: also doubles; complete when a is a
//                                                  square and d is not -- the reference's results
//                                                  on the exceptional inputs of other curves
//                                                  are whatever these formulas give, here too)
//   dbl   dbl-2008-bbjlp   3M + 4S + 1a
// Field: FpMontRT (fp_rt.h), a and d from the curve's parameter block.  Ladder: signed 4-bit
// fixed windows over the table P..8P, as edwards.h / ladder.h.
#pragma once

#include "fp_rt.h"
#include "ladder.h"

namespace ell {

struct EdcWork {
  typedef FpMontRT F;
  typedef F::El El;
  static constexpr int L = 8;
  static constexpr int BYTES = 32;
  static constexpr int NNIB = 64;
  static constexpr int NWIN = 65;              // 64 signed windows + the carry window

  struct P {
    El X, Y, Z;
  };

  ELL_HD static El ca() { return F::curve_a(); }
  ELL_HD static El cd() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = ELL_RT.d_m[i];
    return r;
  }
  ELL_HD static P identity() {
    P r; r.X = F::zero(); r.Y = F::one(); r.Z = F::one(); return r;
  }
  ELL_HD static P select(bool c, const P& x, const P& y) {
    P r;
    r.X = fe_select<F>(c, x.X, y.X);
    r.Y = fe_select<F>(c, x.Y, y.Y);
    r.Z = fe_select<F>(c, x.Z, y.Z);
    return r;
  }
  ELL_HD static P cneg(const P& p, bool neg) {
    P r = p;
    r.X = fe_select<F>(neg, F::neg(p.X), p.X);
    return r;
  }
  ELL_HD static P dbl(const P& p) {
    El B = F::sqr(F::add(p.X, p.Y));
    El C = F::sqr(p.X);
    El D = F::sqr(p.Y);
    El E = F::mul(ca(), C);
    El Ff = F::add(E, D);
    El H = F::sqr(p.Z);
    El J = F::sub(F::sub(Ff, H), H);
    P r;
    r.X = F::mul(F::sub(F::sub(B, C), D), J);
    r.Y = F::mul(Ff, F::sub(E, D));
    r.Z = F::mul(Ff, J);
    return r;
  }
  ELL_HD static P add(const P& p, const P& q, bool do_add = true) {
    El A = F::mul(p.Z, q.Z);
    El B = F::sqr(A);
    El C = F::mul(p.X, q.X);
    El D = F::mul(p.Y, q.Y);
    El E = F::mul(F::mul(cd(), C), D);
    El Ff = F::sub(B, E);
    El G = F::add(B, E);
    El t = F::sub(F::sub(F::mul(F::add(p.X, p.Y), F::add(q.X, q.Y)), C), D);
    P r;
    r.X = F::mul(F::mul(A, Ff), t);
    r.Y = F::mul(F::mul(A, G), F::sub(D, F::mul(ca(), C)));
    r.Z = F::mul(Ff, G);
    return select(do_add, r, p);
  }

  ELL_HD static El load_fe(const u8* p) {
    u32 t[8];
    load_be<8>(t, p, 32);
    return F::from_plain(t);
  }
  ELL_HD static P load_affine(const u8* xy, size_t i) {
    P r;
    r.X = load_fe(xy + i * 64);
    r.Y = load_fe(xy + i * 64 + 32);
    r.Z = F::one();
    return r;
  }
  ELL_HD static void store_proj(u32* out, size_t n, size_t i, const P& p) {
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {
      out[(size_t)(0 * 8 + l) * n + i] = p.X.v[l];
      out[(size_t)(1 * 8 + l) * n + i] = p.Y.v[l];
      out[(size_t)(2 * 8 + l) * n + i] = p.Z.v[l];
    }
  }

  // tbl[j-1] = j*P, j = 1..8
  ELL_HD static void build_table8(P* tbl, const P& p) {
    tbl[0] = p;
    ELL_NOUNROLL
    for (int j = 2; j <= 8; j++) {
      P t;
      if (j & 1) t = add(tbl[j - 2], p);
      else t = dbl(tbl[j / 2 - 1]);
      tbl[j - 1] = t;
    }
  }
  template <int NS>
  ELL_HD static P run_w4(const DigitStore& ds, const P* tbl) {
    P acc = identity();
    ELL_NOUNROLL
    for (int w = NWIN - 1; w >= 0; w--) {
      if (w != NWIN - 1) {
        ELL_NOUNROLL
        for (int j = 0; j < 4; j++) acc = dbl(acc);
      }
      ELL_NOUNROLL
      for (int s = 0; s < NS; s++) {
        int d = ds.get(w * NS + s);
        int ad = d < 0 ? -d : d;
        int e = ad ? ad - 1 : 0;
        acc = add(acc, cneg(tbl[s * 8 + e], d < 0), ad != 0);
      }
    }
    return acc;
  }

  // ---- work items: Point#mul (edwards.js:362-367 -> base.js:86-126), mulAdd, Point#add -------
  ELL_HD static void mul_var(size_t i, size_t n, const u8* ks, const u8* xy, P* tbl_all,
                             const DigitStore& ds, u32* out) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    P* tbl = tbl_all + i * 8;
    build_table8(tbl, load_affine(xy, i));
    recode_w4<8, NNIB, true>(k, ds, 0, 1);
    store_proj(out, n, i, run_w4<1>(ds, tbl));
  }
  ELL_HD static void mul_add2(size_t i, size_t n, const u8* k1s, const u8* xy1, const u8* k2s,
                              const u8* xy2, P* tbl_all, const DigitStore& ds, u32* out) {
    u32 k1[8], k2[8];
    load_be<8>(k1, k1s + i * 32, 32);
    load_be<8>(k2, k2s + i * 32, 32);
    P* tbl = tbl_all + i * 16;
    build_table8(tbl, load_affine(xy1, i));
    build_table8(tbl + 8, load_affine(xy2, i));
    recode_w4<8, NNIB, true>(k1, ds, 0, 2);
    recode_w4<8, NNIB, true>(k2, ds, 1, 2);
    store_proj(out, n, i, run_w4<2>(ds, tbl));
  }
  // a x^2 + y^2 == 1 + d x^2 y^2 (EdwardsCurve#validate, edwards.js:99-112, c = 1)
  ELL_HD static bool on_curve(const El& x, const El& y) {
    El x2 = F::sqr(x), y2 = F::sqr(y);
    El lhs = F::add(F::mul(ca(), x2), y2);
    El rhs = F::add(F::one(), F::mul(F::mul(cd(), x2), y2));
    return F::eq(lhs, rhs);
  }
  // operands that are not on the curve are outside the engine's domain (edwards.h domain_mark)
  ELL_HD static void domain_mark(size_t i, const u8* xy1, const u8* xy2, u8* out_xy, u8* out_inf) {
    bool on = true;
    if (xy1) on = on_curve(load_fe(xy1 + i * 64), load_fe(xy1 + i * 64 + 32));
    if (xy2) on = on && on_curve(load_fe(xy2 + i * 64), load_fe(xy2 + i * 64 + 32));
    if (on) return;
    out_inf[i] = 2;
    if (out_xy) {
      ELL_NOUNROLL
      for (int b = 0; b < 64; b++) out_xy[i * 64 + b] = 0;
    }
  }
  // a set inf flag stands for the identity (0, 1)
  ELL_HD static void point_add(size_t i, size_t n, const u8* xy1, const u8* inf1, const u8* xy2,
                               const u8* inf2, u32* out) {
    P p = load_affine(xy1, i);
    P q = load_affine(xy2, i);
    p = select(inf1 && inf1[i], identity(), p);
    q = select(inf2 && inf2[i], identity(), q);
    store_proj(out, n, i, add(p, q));
  }
  // Point#normalize (edwards.js:377-390): x = X / Z, y = Y / Z, one inversion per K items
  ELL_HD static void normalize(size_t t, size_t T, size_t n, int K, const u32* proj, u32* pre,
                               u8* out_xy, u8* out_inf) {
    El acc = F::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      El z;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) z.v[l] = proj[(size_t)(2 * 8 + l) * n + i];
      z = fe_select<F>(F::is_zero(z), F::one(), z);
      ELL_UNROLL
      for (int l = 0; l < 8; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = F::mul(acc, z);
    }
    El inv = F::inv(acc);
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      El X, Y, Z, pr;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        X.v[l] = proj[(size_t)(0 * 8 + l) * n + i];
        Y.v[l] = proj[(size_t)(1 * 8 + l) * n + i];
        Z.v[l] = proj[(size_t)(2 * 8 + l) * n + i];
        pr.v[l] = pre[(size_t)l * n + i];
      }
      bool zz = F::is_zero(Z);                      // only off the curve / on incomplete curves
      El z = fe_select<F>(zz, F::one(), Z);
      El zinv = F::mul(inv, pr);
      inv = F::mul(inv, z);
      El x = F::mul(X, zinv);
      El y = F::mul(Y, zinv);
      if (zz) { x = F::zero(); y = F::zero(); }
      u32 w[8];
      F::to_plain(w, x);
      store_be<8>(out_xy + i * 64, w, 32);
      F::to_plain(w, y);
      store_be<8>(out_xy + i * 64 + 32, w, 32);
      if (out_inf) out_inf[i] = 0;
    }
  }
};

}  // namespace ell
