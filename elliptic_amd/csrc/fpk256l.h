// ellgpu -- secp256k1 base field in 9 signed limbs of 29 bits ("lazy" representation).
//
// Why a second representation (profiles/r02_u29_probe.log): on gfx950 every
// VALU instruction of this instruction mix costs a wave ~4.3 cycles of SIMD time, a carry
// operation as much as a multiply.  With 32-bit saturated limbs (FpK256, fp.h) a field multiply
// is 72 v_mad_u64_u32 + ~80 carry-class instructions + ~25 moves, and every field addition is a
// 13-instruction carry chain.  With 29-bit limbs in 32-bit words nine partial products fit a
// 64-bit accumulator without any carry handling (102 v_mad_i64_i32 + ~40 shift/mask
// instructions per multiply, measured 0.88x the saturated multiply at 3 waves/SIMD), additions
// and subtractions are nine independent 32-bit operations, and sums of products share one
// reduction (mul2).
//
// Representation.  x = sum v[i] * 2^(29 i), v[i] signed 32-bit.  Every value the GENERIC
// interface (add, sub, neg, dbl, mul_pow2, mul, sqr, from_plain, ...) returns is in N form:
//     |v[i]| <= 2^29 + 2^18  (i < 8),   -2^4 <= v[8] < 2^25 + 2^4,   -2^240 < x < 2^257 + 2^240 (< 4p)
// (direct mul / sqr / mul2 outputs are tighter: v[0], v[1], v[3..7] in [0, 2^29), v[2] in
// (-2^24, 2^29 + 2^18), v[8] in [0, 2^24), -2^83 < x < 2^256 + 2^76 < 2p)
// -- NOT canonical: is_zero / eq / is_odd / to_plain canonicalise first (cold paths only), and
// is_zero_w() is the cheap exact test for a direct mul / sqr / mul2 output (whose limbs are
// non-negative).  The *_l primitives are lazy (no normalisation); the hot group-law formulas
// in short.h use them under the bounds stated at each call site:
//     mul / sqr / mul2 operands:  max_k sum_{i+j=k} |a_i| |b_j|  <  2^63 - 2^50
//     (N x N = 9 * 2^58; one operand may be a difference or a double of N values).
// VALUES may be negative: the column sums of signed limbs are signed, and so is the carry that
// folds back at the end of a product -- a product output lies in (-2^83, 2^256 + 2^76), a lazy
// difference a - b + K p a little below zero when K p does not cover b.  Nothing depends on the
// sign except the canonicalisation (to_plain adds 8p first) and the offsets K p, whose only job
// is to keep limb 8 small and positive so that norm's top fold stays exact.
// The host build (tests/hostsim, -DELL_BOUNDS_CHECK) asserts the column bound at run time.
//
// Replaces bn.js `Red` + `K256` (dist/elliptic.js:6888-6931, 7078-7302) for secp256k1 like
// FpK256 does; which of the two CvSecp256k1 uses is ELL_K256_LAZY (curves.h).
#pragma once

#include "fp.h"
#include "k256l_asm.h"

#if defined(ELL_BOUNDS_CHECK)
#include <assert.h>
#include <stdio.h>
namespace ell { inline const char*& k256l_where() { static thread_local const char* w = "?"; return w; } }
#define ELL_K256L_AT(x) (ell::k256l_where() = (x))
#else
#define ELL_K256L_AT(x) ((void)0)
#endif

namespace ell {

typedef int32_t i32;

struct FpK256L {
  static constexpr int L = 8;            // 32-bit words of a plain value (scalars, byte I/O)
  static constexpr int NS = 9;           // stored limbs
  static constexpr bool LAZY = true;
  static constexpr bool HAS_SQRT = true;
  typedef Fe<9> El;                      // v[i] holds the two's complement bits of a signed limb
  static constexpr u32 M = (1u << 29) - 1;
  static constexpr i32 R0 = 31264, R1 = 256;       // 2^261 = R1 * 2^29 + R0  (mod p)

  ELL_HD static i32 s(u32 x) { return (i32)x; }
  ELL_HD static void get_p(u32 (&p)[8]) { FpK256::get_p(p); }
  ELL_HD static El zero() { El r; bn_zero<9>(r.v); return r; }
  ELL_HD static El one() { El r; bn_zero<9>(r.v); r.v[0] = 1; return r; }

  // ---- conversions ----------------------------------------------------------------------------
  // any 256-bit value -> exact 29-bit digits (N form; the value may be >= p)
  ELL_HD static El from_plain(const u32 (&a)[8]) {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, w = bit >> 5, sh = bit & 31;
      u32 v = a[w] >> sh;
      if (sh > 3 && w + 1 < 8) v |= a[w + 1] << (32 - sh);
      r.v[i] = v & M;
    }
    return r;
  }
  // canonical residue in [0, p) as eight 32-bit words (cold: outputs, comparisons)
  ELL_HD static void to_plain(u32 (&out)[8], const El& a) {
    // + 8p (any value in (-8p, 8p) becomes positive), then a sequential carry: digits d_0..d_7 in
    // [0, 2^29), the rest in the top word
    i32 d[9];
    i32 c = 0;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) {
      i32 t = s(a.v[i]) + c - (i == 0 ? 977 * 8 : (i == 1 ? 8 * 8 : 0));
      d[i] = (i32)((u32)t & M);
      c = t >> 29;
    }
    d[8] = s(a.v[8]) + c + (8 << 24);                  // > 0, < 2^29
    // pack into nine 32-bit words (the ninth holds bits 256..)
    u32 w[9];
    ELL_UNROLL
    for (int j = 0; j < 9; j++) w[j] = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
      u64 v = (u64)(u32)d[i] << sh;
      w[k] |= (u32)v;
      if (k + 1 < 9) w[k + 1] |= (u32)(v >> 32);
    }
    // fold the ninth word: 2^256 = 2^32 + 977, then two conditional subtractions of p
    u32 t8 = w[8];
    u64 acc = (u64)w[0] + (u64)t8 * 977u;
    u32 r[8];
    r[0] = (u32)acc; acc >>= 32;
    acc += (u64)w[1] + t8; r[1] = (u32)acc; acc >>= 32;
    ELL_UNROLL
    for (int i = 2; i < 8; i++) { acc += w[i]; r[i] = (u32)acc; acc >>= 32; }
    u32 top = (u32)acc;                                  // 0 or 1
    u32 p[8]; get_p(p);
    ELL_NOUNROLL
    for (int it = 0; it < 2; it++) {                     // value < 2^256 + 2^38 < 2p
      u32 sres[8];
      u32 br = bn_sub<8>(sres, r, p);
      bool take = (top != 0) || (br == 0);
      ELL_UNROLL
      for (int i = 0; i < 8; i++) r[i] = take ? sres[i] : r[i];
      top = take ? top - br : top;
    }
    bn_copy<8>(out, r);
  }
  ELL_HD static bool is_zero(const El& a) { u32 t[8]; to_plain(t, a); return bn_is_zero<8>(t); }
  ELL_HD static bool eq(const El& a, const El& b) { u32 x[8], y[8]; to_plain(x, a); to_plain(y, b); return bn_eq<8>(x, y); }
  ELL_HD static bool is_odd(const El& a) { u32 t[8]; to_plain(t, a); return t[0] & 1; }
  // exact zero test for a DIRECT mul / sqr / mul2 output: its value lies in (-2^83, 2p), so it is
  // 0 (mod p) iff it is 0 or p.  Limbs 0, 1, 3..8 of such an output are non-negative digits below
  // 2^29 and only limb 2 may be (slightly) negative, which cannot cancel against the others (limbs
  // 0 and 1 sum to less than 2^58): the value is 0 iff every limb is 0.  The value p needs limb 8
  // at its maximum: rare enough for the canonical test.
  ELL_HD static bool is_zero_w(const El& a) {
    u32 o = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) o |= a.v[i];
    if (o == 0) return true;
    if (ELL_UNLIKELY(a.v[8] >= (1u << 24) - 2)) return is_zero(a);
    return false;
  }

  // ---- lazy primitives ------------------------------------------------------------------------
  ELL_HD static El add_l(const El& a, const El& b) {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i];
    return r;
  }
  // a - b + K p   (K p = K 2^256 - K 2^32 - 977 K: limb 8 += K 2^24, limb 1 -= 8 K, limb 0 -= 977 K)
  template <int K>
  ELL_HD static El sub_l(const El& a, const El& b) {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i];
    r.v[0] -= 977u * K;
    r.v[1] -= 8u * K;
    r.v[8] += (u32)K << 24;
    return r;
  }
  template <int K>
  ELL_HD static El neg_l(const El& a) { return sub_l<K>(zero(), a); }
  // c ? K p - a : a
  template <int K>
  ELL_HD static El cneg_l(const El& a, bool c) {
    El n = neg_l<K>(a), r;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) r.v[i] = c ? n.v[i] : a.v[i];
    return r;
  }
  // Fold of limb 8 above bit 24 for norm / shl_norm: with hi = a8 >> 24 >= 1 the value is at
  // least hi * 2^256 minus the (bounded) negative part of the lower limbs, so (hi - 1) p can be
  // taken out without the value going negative -- never all of hi p: a difference A - B + K p
  // whose top limbs cancel may be smaller than its top limb suggests.  Limb 8 stays below 2^25.
  ELL_HD static i32 top_fold(i32 a8, i32& r8) {
    i32 hi = a8 >> 24;
    i32 f = hi >= 1 ? hi - 1 : 0;
    r8 = a8 - (f << 24);
    return f;
  }
  // parallel carry pass + top fold: any lazy value with |limbs| < 2^31 and a non-negative
  // value -> N form
  ELL_HD static El norm(const El& a) {
    El r;
    i32 c[9];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) c[i] = s(a.v[i]) >> 29;
    i32 r8;
    i32 f = top_fold(s(a.v[8]), r8);
    r.v[0] = (a.v[0] & M) + (u32)(f * 977);
    r.v[1] = (a.v[1] & M) + (u32)c[0] + (u32)(f * 8);
    ELL_UNROLL
    for (int i = 2; i < 8; i++) r.v[i] = (a.v[i] & M) + (u32)c[i - 1];
    r.v[8] = (u32)(r8 + c[7]);
    return r;
  }
  // (a << K) in N form, a in N form (K <= 3): the shift and the carry pass in one
  template <int K>
  ELL_HD static El shl_norm(const El& a) {
    El r;
    i32 r8;
    i32 f = top_fold(s(a.v[8]) << K, r8);
    r.v[0] = ((a.v[0] << K) & M) + (u32)(f * 977);
    r.v[1] = ((a.v[1] << K) & M) + (u32)(s(a.v[0]) >> (29 - K)) + (u32)(f * 8);
    ELL_UNROLL
    for (int i = 2; i < 8; i++) r.v[i] = ((a.v[i] << K) & M) + (u32)(s(a.v[i - 1]) >> (29 - K));
    r.v[8] = (u32)(r8 + (s(a.v[7]) >> (29 - K)));
    return r;
  }
  // a / 2 mod p for a lazy value a with |limbs| < 2^30: (a + (a odd ? p : 0)) >> 1, limb-wise
  // (floor semantics, so signed limbs are fine), the bit dropping out of limb i+1 entering limb i
  // at 2^28.  Limbs of the result: |.| < 2^29 + 2^28.
  ELL_HD static El half_l(const El& a) {
    const u32 odd = 0u - (a.v[0] & 1u);                // the value's parity is limb 0's
    // p = (2^29 - 977, 2^29 - 9, 2^29 - 1 x 6, 2^24 - 1)
    u32 t[9];
    t[0] = a.v[0] + (odd & ((1u << 29) - 977u));
    t[1] = a.v[1] + (odd & ((1u << 29) - 9u));
    ELL_UNROLL
    for (int i = 2; i < 8; i++) t[i] = a.v[i] + (odd & M);
    t[8] = a.v[8] + (odd & ((1u << 24) - 1u));
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = (u32)(s(t[i]) >> 1) + ((t[i + 1] & 1u) << 28);
    r.v[8] = (u32)(s(t[8]) >> 1);
    return r;
  }

  // ---- products ---------------------------------------------------------------------------------
#if defined(ELL_BOUNDS_CHECK)
  static void check_operands(const El& a, const El& b, __int128 (&col)[17]) {
    for (int k = 0; k < 17; k++) {
      __int128 sum = 0;
      for (int i = 0; i < 9; i++) {
        int j = k - i;
        if (j < 0 || j > 8) continue;
        __int128 x = s(a.v[i]), y = s(b.v[j]);
        sum += (x < 0 ? -x : x) * (y < 0 ? -y : y);
      }
      col[k] += sum;
    }
  }
  static void check_cols(const __int128 (&col)[17]) {
    const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
    for (int k = 0; k < 17; k++) assert(col[k] < lim && "fpk256l: column sum exceeds 63 bits");
  }
  static void dump(const char* what, const El& a) {
    fprintf(stderr, "[%s] %s:", k256l_where(), what);
    for (int i = 0; i < 9; i++) fprintf(stderr, " %d", s(a.v[i]));
    fprintf(stderr, "\n");
  }
#endif
  // tail shared by all products: limbs r[0..8] in [0, 2^29), carry c out of limb 8, h8 the last
  // limb of the high half; limb 9 (weight 2^261) and the bits of limb 8 above 2^24 fold into
  // limbs 0..2, which stay within 2^29 + 2^18
  ELL_HD static El tail(const i32 (&r)[9], i64 c, i32 h8) {
    El o;
    c += (i64)h8 * R1;
    i32 g = (i32)((u32)c & M), g2 = (i32)(c >> 29);
    i32 hi = r[8] >> 24;                               // 0..31
    i64 t = (i64)r[0] + (i64)g * R0 + (i64)(hi * 977);
    o.v[0] = (u32)t & M; t >>= 29;
    t += (i64)r[1] + (i64)g * R1 + (i64)g2 * R0 + (i64)(hi * 8);
    o.v[1] = (u32)t & M; t >>= 29;
    o.v[2] = (u32)(r[2] + (i32)t + g2 * R1);
    ELL_UNROLL
    for (int i = 3; i < 8; i++) o.v[i] = (u32)r[i];
    o.v[8] = (u32)(r[8] & ((1 << 24) - 1));
    return o;
  }
  // portable column code (host passes; the device uses the generated asm)
  template <int NP>
  ELL_HD static El mul_generic(const El* const (&x)[NP], const El* const (&y)[NP]) {
    i32 r[9];
    i64 d = 0, c = 0;
    ELL_UNROLL
    for (int q = 0; q < NP; q++) {
      ELL_UNROLL
      for (int i = 1; i <= 8; i++) d += (i64)s(x[q]->v[i]) * s(y[q]->v[9 - i]);
    }
    i32 h = (i32)((u32)d & M), hp = 0;
    d >>= 29;
    ELL_UNROLL
    for (int k = 0; k <= 8; k++) {
      if (k >= 1) {
        if (k < 8) {
          ELL_UNROLL
          for (int q = 0; q < NP; q++) {
            ELL_UNROLL
            for (int i = k + 1; i <= 8; i++) d += (i64)s(x[q]->v[i]) * s(y[q]->v[9 + k - i]);
          }
          hp = h; h = (i32)((u32)d & M); d >>= 29;
        } else {
          hp = h; h = (i32)d;
        }
      }
      ELL_UNROLL
      for (int q = 0; q < NP; q++) {
        ELL_UNROLL
        for (int i = 0; i <= k; i++) c += (i64)s(x[q]->v[i]) * s(y[q]->v[k - i]);
      }
      if (k >= 1) c += (i64)hp * R1;
      c += (i64)h * R0;
      r[k] = (i32)((u32)c & M);
      c >>= 29;
    }
    return tail(r, c, h);
  }
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    { __int128 col[17] = {0}; check_operands(a, b, col);
      const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
      for (int k = 0; k < 17; k++) if (col[k] >= lim) { dump("mul a", a); dump("mul b", b); }
      check_cols(col); }
#endif
#if defined(ELL_HAVE_K256L_ASM)
    i32 r[9]; i64 c; i32 h8;
    k256l::k256l_mul_cols(r, c, h8, (const i32(&)[9])a.v, (const i32(&)[9])b.v);
    return tail(r, c, h8);
#else
    const El* const x[1] = {&a};
    const El* const y[1] = {&b};
    return mul_generic<1>(x, y);
#endif
  }
  ELL_HD static El sqr(const El& a) {
#if defined(ELL_BOUNDS_CHECK)
    { __int128 col[17] = {0}; check_operands(a, a, col);
      const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
      for (int k = 0; k < 17; k++) if (col[k] >= lim) { dump("sqr a", a); }
      check_cols(col); }
#endif
#if defined(ELL_HAVE_K256L_ASM)
    i32 r[9]; i64 c; i32 h8;
    k256l::k256l_sqr_cols(r, c, h8, (const i32(&)[9])a.v);
    return tail(r, c, h8);
#else
    const El* const x[1] = {&a};
    const El* const y[1] = {&a};
    return mul_generic<1>(x, y);
#endif
  }
  // a * b + e * f with one reduction
  ELL_HD static El mul2(const El& a, const El& b, const El& e, const El& f) {
#if defined(ELL_BOUNDS_CHECK)
    { __int128 col[17] = {0}; check_operands(a, b, col); check_operands(e, f, col);
      const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
      for (int k = 0; k < 17; k++) if (col[k] >= lim) { dump("mul2 a", a); dump("mul2 b", b); dump("mul2 e", e); dump("mul2 f", f); }
      check_cols(col); }
#endif
#if defined(ELL_HAVE_K256L_ASM)
    i32 r[9]; i64 c; i32 h8;
    k256l::k256l_mul2_cols(r, c, h8, (const i32(&)[9])a.v, (const i32(&)[9])b.v, (const i32(&)[9])e.v, (const i32(&)[9])f.v);
    return tail(r, c, h8);
#else
    const El* const x[2] = {&a, &e};
    const El* const y[2] = {&b, &f};
    return mul_generic<2>(x, y);
#endif
  }

  // ---- generic interface: N form in, N form out ---------------------------------------------------
  ELL_HD static El add(const El& a, const El& b) { return norm(add_l(a, b)); }
  ELL_HD static El sub(const El& a, const El& b) { return norm(sub_l<4>(a, b)); }       // N values are < 2^257 + eps < 4p
  ELL_HD static El neg(const El& a) { return norm(neg_l<4>(a)); }
  ELL_HD static El dbl(const El& a) { return shl_norm<1>(a); }
  template <int K>
  ELL_HD static El mul_pow2(const El& a) { return shl_norm<K>(a); }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  // a^-1 (0 for 0): canonical words -> the division steps of the saturated field -> back
  static ELL_HD_NOINLINE El inv(const El& a) {
    FpK256::El t;
    to_plain(t.v, a);
    FpK256::El r = FpK256::inv(t);
    return from_plain(r.v);
  }
  // a^((p+1)/4), the chain of FpK256::sqrt (bn.js Red#sqrt takes the same power)
  static ELL_HD_NOINLINE El sqrt(const El& a) {
    El x2 = mul(sqr(a), a);
    El x3 = mul(sqr(x2), a);
    El x6 = mul(sqr_n(x3, 3), x3);
    El x9 = mul(sqr_n(x6, 3), x3);
    El x11 = mul(sqr_n(x9, 2), x2);
    El x22 = mul(sqr_n(x11, 11), x11);
    El x44 = mul(sqr_n(x22, 22), x22);
    El x88 = mul(sqr_n(x44, 44), x44);
    El x176 = mul(sqr_n(x88, 88), x88);
    El x220 = mul(sqr_n(x176, 44), x44);
    El x223 = mul(sqr_n(x220, 3), x3);
    El t = mul(sqr_n(x223, 23), x22);
    t = mul(sqr_n(t, 6), x2);
    return sqr_n(t, 2);
  }
};

}  // namespace ell
