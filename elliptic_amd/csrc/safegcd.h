// ellgpu -- modular inversion by Bernstein-Yang division steps ("safegcd", Bernstein & Yang,
// "Fast constant-time gcd computation and modular inversion", 2019), replacing bn.js `invm`
// (reference: dist/elliptic.js:7031-7050 `Red#invm` -> `BN#_invmp`, an extended binary gcd).
// Only the result -- the unique inverse in [0, m), 0 for 0 -- is part of the contract.
//
// Shape: a fixed number of rounds (no data-dependent control flow, every lane of a wave does
// the same work); each round runs 30 division steps on the low words of f and g with plain
// 32-bit adds / shifts / masks, collects them in a 2x2 transition matrix, and applies the
// matrix to the full-width (f, g) and, modulo m, to (d, e) with signed 30-bit limbs and
// 64-bit multiply-accumulates -- no carry flags anywhere.  floor((49 d + 57) / 17) division
// steps suffice for a d-bit modulus (Theorem 11.2); the constants come from
// tools/gen_consts.py.  About a fifth of the cost of the Fermat power it replaces.
#pragma once

#include <stdint.h>

#include "common.h"

namespace ell {

typedef int32_t i32;
typedef int64_t i64;

template <class MP>
struct SafeGcd {
  static constexpr int L = MP::L;               // 32-bit limbs of the modulus
  static constexpr int NL = MP::SG_NL;          // signed 30-bit limbs: 30 NL >= bits + 2
  static constexpr int ROUNDS = MP::SG_ROUNDS;
  static constexpr u32 M30 = 0x3FFFFFFFu;

  // 30 division steps on the low words; (u v; q r) = 2^30 times the transition matrix
  ELL_HD static i32 divsteps30(i32 eta, u32 f, u32 g, i32& uo, i32& vo, i32& qo, i32& ro) {
    u32 u = 1, v = 0, q = 0, r = 1;
    ELL_UNROLL
    for (int i = 0; i < 30; i++) {
      u32 c1 = (u32)(eta >> 31);                // all ones when delta > 0
      u32 c2 = 0u - (g & 1u);                   // all ones when g is odd
      u32 x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;     // conditionally negated f, u, v
      g += x & c2; q += y & c2; r += z & c2;
      c1 &= c2;                                 // swap?
      eta = (eta ^ (i32)c1) - ((i32)c1 + 1);
      f += g & c1; u += q & c1; v += r & c1;
      g >>= 1; u <<= 1; v <<= 1;
    }
    uo = (i32)u; vo = (i32)v; qo = (i32)q; ro = (i32)r;
    return eta;
  }

  // (f, g) <- (u f + v g, q f + r g) / 2^30   (exact)
  ELL_HD static void update_fg(i32 (&f)[NL], i32 (&g)[NL], i32 u, i32 v, i32 q, i32 r) {
    i64 cf = (i64)u * f[0] + (i64)v * g[0];
    i64 cg = (i64)q * f[0] + (i64)r * g[0];
    cf >>= 30; cg >>= 30;
    ELL_UNROLL
    for (int i = 1; i < NL; i++) {
      cf += (i64)u * f[i] + (i64)v * g[i];
      cg += (i64)q * f[i] + (i64)r * g[i];
      f[i - 1] = (i32)((u32)cf & M30); cf >>= 30;
      g[i - 1] = (i32)((u32)cg & M30); cg >>= 30;
    }
    f[NL - 1] = (i32)cf;
    g[NL - 1] = (i32)cg;
  }

  // (d, e) <- (u d + v e, q d + r e) / 2^30 mod m, kept in (-2m, m)
  ELL_HD static void update_de(i32 (&d)[NL], i32 (&e)[NL], i32 u, i32 v, i32 q, i32 r) {
    i32 sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
    i32 md = (u & sd) + (v & se);
    i32 me = (q & sd) + (r & se);
    i64 cd = (i64)u * d[0] + (i64)v * e[0];
    i64 ce = (i64)q * d[0] + (i64)r * e[0];
    // multiples of m that clear the low 30 bits
    md -= (i32)((MP::sg_minv30 * (u32)cd + (u32)md) & M30);
    me -= (i32)((MP::sg_minv30 * (u32)ce + (u32)me) & M30);
    cd += (i64)(i32)MP::sg_m30[0] * md;
    ce += (i64)(i32)MP::sg_m30[0] * me;
    cd >>= 30; ce >>= 30;
    ELL_UNROLL
    for (int i = 1; i < NL; i++) {
      cd += (i64)u * d[i] + (i64)v * e[i] + (i64)(i32)MP::sg_m30[i] * md;
      ce += (i64)q * d[i] + (i64)r * e[i] + (i64)(i32)MP::sg_m30[i] * me;
      d[i - 1] = (i32)((u32)cd & M30); cd >>= 30;
      e[i - 1] = (i32)((u32)ce & M30); ce >>= 30;
    }
    d[NL - 1] = (i32)cd;
    e[NL - 1] = (i32)ce;
  }

  // r = x^-1 mod m for 0 <= x < m given as L 32-bit limbs; r = 0 for x = 0
  ELL_HD static void inv(u32 (&out)[L], const u32 (&x)[L]) {
    i32 f[NL], g[NL], d[NL], e[NL];
    ELL_UNROLL
    for (int i = 0; i < NL; i++) {
      f[i] = (i32)MP::sg_m30[i];
      // bits [30 i, 30 i + 30) of x
      const int w = (30 * i) >> 5, sh = (30 * i) & 31;
      u32 lo = w < L ? x[w < L ? w : 0] : 0u;
      u32 hi = w + 1 < L ? x[w + 1 < L ? w + 1 : 0] : 0u;
      u32 val = sh == 0 ? lo : ((lo >> sh) | (sh > 2 ? hi << (32 - sh) : 0u));
      g[i] = (i32)(val & M30);
      d[i] = 0;
      e[i] = i == 0 ? 1 : 0;
    }
    i32 eta = -1;
    ELL_NOUNROLL
    for (int it = 0; it < ROUNDS; it++) {
      i32 u, v, q, r;
      u32 f0 = (u32)f[0] | ((u32)f[1] << 30);
      u32 g0 = (u32)g[0] | ((u32)g[1] << 30);
      eta = divsteps30(eta, f0, g0, u, v, q, r);
      update_de(d, e, u, v, q, r);
      update_fg(f, g, u, v, q, r);
    }
    // now g = 0, f = +-1 (+-m for x = 0): x^-1 = sign(f) * d.  Bring sign(f) * d, which lies in
    // (-2m, 2m), to [0, 4m) by adding 2m, repack to 32-bit limbs, subtract 2m and m as needed.
    i64 s = (i64)(f[NL - 1] >> 31 | 1);          // +1 or -1
    u32 t30[NL];
    i64 c = 0;
    ELL_UNROLL
    for (int i = 0; i < NL; i++) {
      c += s * d[i] + 2 * (i64)(i32)MP::sg_m30[i];
      t30[i] = (u32)c & M30;
      c >>= 30;
    }
    // c == 0 here: the value is in [0, 4m) and 30 NL >= bits + 2
    u32 t[L + 1];
    ELL_UNROLL
    for (int j = 0; j <= L; j++) {
      // bits [32 j, 32 j + 32) of sum t30[i] 2^(30 i)
      const int i0 = (32 * j) / 30, sh = (32 * j) - 30 * i0;
      u32 a = i0 < NL ? t30[i0 < NL ? i0 : 0] : 0u;
      u32 b = i0 + 1 < NL ? t30[i0 + 1 < NL ? i0 + 1 : 0] : 0u;
      u32 cc = i0 + 2 < NL ? t30[i0 + 2 < NL ? i0 + 2 : 0] : 0u;
      u32 val = a >> sh;
      val |= b << (30 - sh);
      if (60 - sh < 32) val |= cc << (60 - sh);
      t[j] = val;
    }
    u32 m1[L + 1], m2[L + 1];
    ELL_UNROLL
    for (int j = 0; j < L; j++) m1[j] = MP::p[j];
    m1[L] = 0;
    m2[0] = m1[0] << 1;
    ELL_UNROLL
    for (int j = 1; j <= L; j++) m2[j] = (m1[j] << 1) | (m1[j - 1] >> 31);
    u32 s2[L + 1];
    u32 br = bn_sub<L + 1>(s2, t, m2);
    ELL_UNROLL
    for (int j = 0; j <= L; j++) t[j] = br ? t[j] : s2[j];
    br = bn_sub<L + 1>(s2, t, m1);
    ELL_UNROLL
    for (int j = 0; j < L; j++) out[j] = br ? t[j] : s2[j];
  }
};

}  // namespace ell
