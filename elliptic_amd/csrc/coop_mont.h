// ellgpu -- the lanes-per-item layer (coop.h) for the NIST primes up to 256 bits: a Montgomery
// field over nine signed 29-bit limbs, one limb per lane of a 16-lane row, R = 2^261.
//
// Same idea as coop.h's FpK256C -- a product's partial-product rows are nine multiply-adds of the
// whole row, additions are one instruction -- with a reduction that does not depend on the shape
// of the prime: word-serial REDC.  Column i of the 17-column product (lane i; column 16 on the
// scalar unit) is made divisible by 2^29 by adding m_i * p shifted up i lanes, m_i = column i *
// (-p^-1) mod 2^29 computed on the scalar unit from a v_readlane of the running column; after
// nine steps the low nine columns are zero and columns 9..17 ARE the result (one DPP shift down).
// Signed limbs throughout (no offsets K p anywhere): REDC works on two's complement columns, and
// because R is 2^5 times larger than 2^256 a product of two values below ~1.01 p comes out below
// 1.04 p -- nothing is ever conditionally subtracted on the fast path.  Additions, subtractions
// and small multiples fold the VALUE back with one estimate from the top limb (norm): every value
// the interface returns lies in (-2^(PBITS-23), 2^PBITS + 2^(PBITS-23)), limbs below 2^29 + 2^7.
//
// Used for single calls and small batches on p256 / p224 / p192 (coop_work.h CoopNist): the
// reference's JPoint arithmetic (short.js:532-603, 739-800) over bn.js `Mont` (dist/elliptic.js:
// 7308-7381), one item per wave.  Elements are in MONTGOMERY form inside this layer; the one-lane
// kernels' tables and results are plain words (fp.h FpSolinas), converted at the boundary.
#pragma once

#include "coop.h"
#include "coop_consts.h"

namespace ell {

// MC: consts::COOP_P* (nine 29-bit digits of p, n0, R, R^2); F1: the curve's one-lane field (plain
// words <-> elements, the inversion)
template <class MC, class F1>
struct FpMontC {
  typedef FpK256C R_;                                // the row primitives (lane index, DPP moves, readlane)
  static constexpr int CL = ELL_COOP_LANES;
  static constexpr int L = F1::L;                    // 32-bit words of a plain value
  static constexpr bool HAS_SQRT = false;
  typedef Fe<CL> El;
  typedef FpK256C::W64 W64;
  static constexpr u32 M = (1u << 29) - 1;
  static constexpr int TL = (MC::PBITS - 1) / 29;    // the limb that holds bit PBITS - 1 ...
  static constexpr int TB = MC::PBITS - 29 * TL;     // ... and how many bits of it belong to a value below 2^PBITS

  ELL_HD static i32 s(u32 x) { return (i32)x; }
  template <class Fn> ELL_HD static El each(const Fn& f) { return R_::each(f); }
  template <int N> ELL_HD static El up(const El& x) { return R_::template up<N>(x); }
  template <int N> ELL_HD static El down(const El& x) { return R_::template down<N>(x); }
  ELL_HD static i32 at(const El& x, int l) { return R_::at(x, l); }
  ELL_HD static int lane_of(int t) { return R_::lane_of(t); }

  ELL_HD static El c_p() { return by_lane<0>(MC::p29); }
  // digits of a compile-time table by lane: a select chain (an indexed read would be a memory table)
  template <int I, class Tab>
  ELL_HD static El by_lane(const Tab& tab) {
    return each([&](int l) {
      i32 v = 0;
      ELL_UNROLL
      for (int j = 0; j < 9; j++) v = (l == j + I) ? (i32)tab[j] : v;
      return v;
    });
  }
  template <int I>
  ELL_HD static El c_pshift() { return by_lane<I>(MC::p29); }
  ELL_HD static El c_live() { return R_::c_live(); }
  ELL_HD static El c_mask() { return R_::c_mask(); }
  ELL_HD static El zero() { return each([](int) { return 0; }); }
  ELL_HD static El one() { return by_lane<0>(MC::one29); }     // R mod p
  ELL_HD static El c_rr() { return by_lane<0>(MC::rr29); }      // R^2 mod p
  ELL_HD static El plain_one() { return R_::one(); }

  // ---- lazy limb-wise forms -----------------------------------------------------------------------
  ELL_HD static El add_l(const El& a, const El& b) {
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = a.v[t] + b.v[t];
    return r;
  }
  ELL_HD static El sub_l(const El& a, const El& b) {
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = a.v[t] - b.v[t];
    return r;
  }
  // Carry pass + value fold: a lazy value with |limbs| < 2^30 and |value| < 3 * 2^PBITS comes back
  // with limbs in (-2^7, 2^29 + 2^7) and its value in (-2^(PBITS-23), 2^PBITS + 2^(PBITS-23)).
  // The fold subtracts k p for k = floor(value / 2^PBITS) as limb TL shows it (limbs above TL are
  // carry-sized), k in [-2, 2]: |limb - k p_l| stays below 2^31.
  ELL_HD static El norm(const El& a) {
    const El pv = c_p(), mk = c_mask(), live = c_live();
    i64 top = (i64)at(a, TL);
    ELL_UNROLL
    for (int j = TL + 1; j <= 8; j++) top += (i64)at(a, j) << (29 * (j - TL));        // (carry-sized limbs above TL)
    const i64 k64 = top >> TB;
    const i32 k = k64 < -2 ? -2 : (k64 > 2 ? 2 : (i32)k64);
#if defined(ELL_BOUNDS_CHECK)
    for (int t = 0; t < CL; t++) {
      const i64 dd = (i64)s(a.v[t]) - (i64)k * (i64)s(pv.v[t]);
      if (dd >= ((i64)1 << 31) || dd < -((i64)1 << 31)) { fprintf(stderr, "fpmontc norm: limb %d leaves 32 bits\n", t); assert(0); }
    }
#endif
    El d, c;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) d.v[t] = a.v[t] - (u32)k * pv.v[t];
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c.v[t] = (u32)(s(d.v[t]) >> 29);
    const El cin = up<1>(c);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = ((d.v[t] & mk.v[t]) + cin.v[t]) & live.v[t];
    return r;
  }

  // ---- products: nine multiply-adds of the row, then word-serial REDC -----------------------------
#if defined(ELL_BOUNDS_CHECK)
  static void check(const El& a, const El& b, const char* what) {
    __int128 col[17] = {0};
    FpK256L::check_operands(R_::gather(a), R_::gather(b), col);
    // + the reduction's own rows: nine m_i p, |m_i| < 2^29
    const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
    for (int k = 0; k < 17; k++) {
      __int128 red = 0;
      for (int i = 0; i <= 8; i++) { int j = k - i; if (j >= 0 && j <= 8) red += ((__int128)1 << 29) * MC::p29[j]; }
      if (col[k] + red >= lim) { fprintf(stderr, "fpmontc %s: column %d exceeds 63 bits\n", what, k); assert(0); }
    }
    for (int t = 9; t < CL; t++) assert(a.v[t] == 0 && b.v[t] == 0 && "fpmontc: dead lane not zero");
  }
#endif
  // one step of the reduction: column I made divisible by 2^29 (RW: every row reduces a product of
  // its own -- coop.h ln -- and m, the carry and column 16 are per-row values)
  template <bool RW, int I>
  ELL_HD static void redc_step(W64& acc, i64& col16, i64& carry) {
    // p shifted up I lanes, as a per-lane CONSTANT (a DPP move is a convergent operation the
    // compiler will not hoist out of the ladder's loops; a select chain on the lane index it does)
    const El ps = c_pshift<I>();
    const i64 p0 = (i64)MC::p29[0], p8s = (i64)MC::p29[8];
    const i64 v = R_::template ln64<RW, I>(acc) + carry;            // column I as it stands
    const i32 m = (i32)(((u32)v * MC::n0) & M);                    // v + m p_0 = 0 (mod 2^29)
    ELL_UNROLL
    for (int t = 0; t < CL; t++) acc.w[t] += (i64)m * (i64)s(ps.v[t]);
    if (I == 8) col16 += (i64)m * p8s;                             // (lane 16 is off the row)
    // p = -1 (mod 2^29) (p256, p192): m is v's low digit and v + m (2^29 - 1) = ((v >> 29) + m) 2^29
    if constexpr (MC::n0 == 1u && MC::p29[0] == (int)M) carry = (v >> 29) + (i64)m;
    else carry = (v + (i64)m * p0) >> 29;
  }
  template <bool RW = false>
  ELL_HD static El redc(W64 acc, i64 col16) {
    const El live = c_live();
    i64 carry = 0;
    redc_step<RW, 0>(acc, col16, carry);
    redc_step<RW, 1>(acc, col16, carry);
    redc_step<RW, 2>(acc, col16, carry);
    redc_step<RW, 3>(acc, col16, carry);
    redc_step<RW, 4>(acc, col16, carry);
    redc_step<RW, 5>(acc, col16, carry);
    redc_step<RW, 6>(acc, col16, carry);
    redc_step<RW, 7>(acc, col16, carry);
    redc_step<RW, 8>(acc, col16, carry);
    // columns 9..15 (lanes 9..15), 16 (scalar) and the carry into column 9 are the result's limbs 0..7
    W64 res;
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)acc.w[0], 0x100 + 9, 0xF, 0xF, true);
      const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((u64)acc.w[0] >> 32), 0x100 + 9, 0xF, 0xF, true);
      i64 x = (i64)(((u64)hi << 32) | lo);
      const int l = lane_of(0);
      x = l == 7 ? col16 : x;
      x += l == 0 ? carry : 0;
      res.w[0] = x;
    }
#else
    for (int t = 0; t < CL; t++) res.w[t] = t + 9 < CL ? acc.w[t + 9] : 0;
    res.w[7] = col16;
    res.w[0] += carry;
#endif
    // two carry passes (64-bit, then 32-bit): limbs below 2^29 + 2^7; limb 8 takes limb 7's carries
    W64 c1;
    El lo1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) { c1.w[t] = res.w[t] >> 29; lo1.v[t] = (u32)res.w[t] & M; }
    const W64 cin1 = R_::template up64<1>(c1);
    W64 v1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v1.w[t] = (i64)lo1.v[t] + cin1.w[t];
    El c2, r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c2.v[t] = lane_of(t) < 8 ? (u32)(i32)(v1.w[t] >> 29) : 0u;
    const El cin2 = up<1>(c2);
    ELL_UNROLL
    for (int t = 0; t < CL; t++)
      r.v[t] = ((lane_of(t) < 8 ? ((u32)v1.w[t] & M) : (u32)v1.w[t]) + cin2.v[t]) & live.v[t];
    return r;
  }
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    check(a, b, "mul");
#endif
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::columns(acc, col16, a, b);
    return redc(acc, col16);
  }
  ELL_HD static El sqr(const El& a) { return mul(a, a); }
  // four products side by side, one per row of the wave (coop.h Q / mulq)
  static constexpr bool QUAD = true;
  typedef FpK256C::Q Q;
  ELL_HD static Q pack2(const El& a, const El& b) { return R_::pack2(a, b); }
  ELL_HD static Q pack3(const El& a, const El& b, const El& c) { return R_::pack3(a, b, c); }
  ELL_HD static void unpack2(const Q& q, El& a, El& b) { R_::unpack2(q, a, b); }
  ELL_HD static void unpack3(const Q& q, El& a, El& b, El& c) { R_::unpack3(q, a, b, c); }
  ELL_HD static Q mulq(const Q& a, const Q& b) {
    Q r;
#if defined(__HIP_DEVICE_COMPILE__)
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::template columns<true>(acc, col16, a.r[0], b.r[0]);
    r.r[0] = redc<true>(acc, col16);
#else
    for (int j = 0; j < R_::QR; j++) r.r[j] = mul(a.r[j], b.r[j]);
#endif
    return r;
  }

  // ---- the field interface of short.h's generic (non-lazy) formulas --------------------------------
  ELL_HD static El add(const El& a, const El& b) { return norm(add_l(a, b)); }
  ELL_HD static El sub(const El& a, const El& b) { return norm(sub_l(a, b)); }
  ELL_HD static El neg(const El& a) { return norm(sub_l(zero(), a)); }
  ELL_HD static El dbl(const El& a) { return norm(add_l(a, a)); }
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    El r = a;
    ELL_UNROLL
    for (int i = 0; i < K; i++) r = norm(add_l(r, r));
    return r;
  }

  // ---- conversions, tests (cold paths: through the scalar unit and the one-lane field) -------------
  // the value's canonical residue in [0, p) as nine exact 29-bit digits -> eight plain words
  ELL_HD static void canon(u32 (&out)[8], const El& a) {
    i64 d[9];
    ELL_UNROLL
    for (int i = 0; i < 9; i++) d[i] = (i64)at(a, i);
    // value in (-p, 2p): + p, sequential carry, then up to two conditional subtractions of p
    i64 c = 0;
    u32 dig[9];
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      i64 t = d[i] + (i64)MC::p29[i] + c;
      dig[i] = (u32)t & M;
      c = t >> 29;
    }
    // (c is 0 here: the value + p lies in (0, 3p) < 2^261)
    ELL_NOUNROLL
    for (int it = 0; it < 2; it++) {
      u32 sub[9];
      i32 br = 0;
      ELL_UNROLL
      for (int i = 0; i < 9; i++) {
        i32 t = (i32)dig[i] - MC::p29[i] + br;
        sub[i] = (u32)t & M;
        br = t >> 29;
      }
      const bool take = br == 0;                       // dig >= p
      ELL_UNROLL
      for (int i = 0; i < 9; i++) dig[i] = take ? sub[i] : dig[i];
    }
    u32 w[9];
    ELL_UNROLL
    for (int j = 0; j < 9; j++) w[j] = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      const int bit = 29 * i, k = bit >> 5, sh = bit & 31;
      const u64 v = (u64)dig[i] << sh;
      w[k] |= (u32)v;
      if (k + 1 < 9) w[k + 1] |= (u32)(v >> 32);
    }
    ELL_UNROLL
    for (int j = 0; j < 8; j++) out[j] = w[j];
  }
  // plain words of the element (out of Montgomery form): REDC of a * 1, canonical
  ELL_HD static void to_plain(u32 (&out)[L], const El& a) {
    u32 w[8];
    canon(w, mul(a, plain_one()));
    ELL_UNROLL
    for (int j = 0; j < L; j++) out[j] = w[j];
  }
  // wave-uniform plain words -> the element: digits times R^2, reduced
  ELL_HD static El from_plain(const u32 (&a)[L]) {
    u32 w[8];
    ELL_UNROLL
    for (int j = 0; j < 8; j++) w[j] = j < L ? a[j] : 0u;
    return mul(R_::scatter(FpK256L::from_plain(w)), c_rr());
  }
  // L plain words in memory (an entry of the one-lane kernels' tables) -> the element
  ELL_HD static El load_words(const u32* w) {
    const El digs = each([&](int l) {
      const int ll = l > 8 ? 8 : l;
      const int bit = 29 * ll;
      int k = bit >> 5;
      const int sh = bit & 31;
      const u32 lo = k < L ? w[k] : 0u;
      const u32 hi = k + 1 < L ? w[k + 1] : 0u;
      const u64 two = (u64)lo | ((u64)hi << 32);
      return l > 8 ? 0 : (i32)((u32)(two >> sh) & M);
    });
    return mul(digs, c_rr());
  }
  // Zero test.  Every value of the interface lies in (-p, 2p): it is 0 (mod p) iff it is 0 or p, and
  // then limb 0's low 29 bits are 0's or p's -- anything else (all but one value in 2^28) is not zero;
  // the canonical digits decide the rest.  (Montgomery form keeps 0 at 0.)
  ELL_HD static bool is_zero(const El& a) {
    const u32 r0 = (u32)at(a, 0) & M;
    if (ELL_UNLIKELY(r0 == 0u || r0 == ((u32)MC::p29[0] & M))) {
      u32 w[8];
      canon(w, a);
      return bn_is_zero<8>(w);
    }
    return false;
  }
  ELL_HD static bool eq(const El& a, const El& b) { return is_zero(sub(a, b)); }
  // a^-1 (0 for 0): through the one-lane field's division steps
  static ELL_HD_NOINLINE El inv(const El& a) {
    u32 w[L];
    to_plain(w, a);
    typename F1::El y = F1::inv(F1::from_plain(w));
    u32 v[L];
    F1::to_plain(v, y);
    return from_plain(v);
  }
};

// ---- the same row, reduced by FOLDING (round 6) ------------------------------------------------------
// For a prime of more than 232 bits (p256) the nine-digit row holds plain residues and a product
// is reduced without the word-serial chain above -- the reduction csrc/coop_wide.h uses for p384 /
// p521: after two carry passes digit 9 + j of the product (lanes 9..15, column 16's two digits) is a
// 29-bit number worth the constant vector fold29[j] = 2^(29 (9 + j)) mod p -- nine independent
// multiply-adds of the row --, then what stands above 2^PBITS (34 bits at most) is worth top29 =
// 2^PBITS mod p and folds once more.  No step waits for a v_readlane -> scalar -> v_mad of the step
// before (FpMontC: 94 VALU + 61 SALU at 2.9 ns each on a lone wave).  Interface, value range
// (-2^(PBITS-23), 2^PBITS + 2^(PBITS-23)), norm, the conversions' canonical digits: FpMontC's.
template <class MC, class F1>
struct FpFoldC : FpMontC<MC, F1> {
  typedef FpMontC<MC, F1> B;
  typedef FpK256C R_;
  typedef typename B::El El;
  typedef typename B::W64 W64;
  typedef FpK256C::Q Q;
  static constexpr int CL = B::CL;
  static constexpr int L = B::L;
  static constexpr u32 M = B::M;
  static constexpr int TL = 8, TB = MC::PBITS - 29 * 8;
  static_assert(MC::PBITS > 232 && MC::PBITS <= 261 - 5, "FpFoldC: the top digit holds bit PBITS - 1 and the fold's overflow");
  static constexpr bool QUAD = true;

  ELL_HD static i32 s(u32 x) { return (i32)x; }
  ELL_HD static int lane_of(int t) { return R_::lane_of(t); }
  template <int J>
  ELL_HD static El c_fold() { return B::template by_lane<0>(MC::fold29[J]); }
  ELL_HD static El c_top() { return B::template by_lane<0>(MC::top29); }
  ELL_HD static El c_top1() { return B::template by_lane<1>(MC::top29); }
  ELL_HD static El one() { return R_::one(); }

  // one carry pass over 64-bit lanes; lane 8 (the top digit) keeps its value whole and hands nothing on
  ELL_HD static W64 carry64(const W64& x) {
    W64 c, lo;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      const bool k = lane_of(t) >= 8;
      c.w[t] = k ? 0 : (x.w[t] >> 29);
      lo.w[t] = k ? x.w[t] : (i64)((u32)x.w[t] & M);
    }
    const W64 cin = R_::template up64<1>(c);
    W64 r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.w[t] = lo.w[t] + cin.w[t];
    return r;
  }
  // columns 0..15 (one per lane) and 16 -> the product's residue (RW: every row reduces a product of its own)
  template <bool RW = false>
  ELL_HD static El reduce(const W64& acc, i64 col16) {
    const El live = B::c_live();
    // the columns as digits: two carry passes, as coop.h FpK256C::tail
    W64 c1;
    El lo1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) { c1.w[t] = acc.w[t] >> 29; lo1.v[t] = (u32)acc.w[t] & M; }
    const W64 cin1 = R_::template up64<1>(c1);
    W64 v1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v1.w[t] = (i64)lo1.v[t] + cin1.w[t];
    col16 += R_::template ln64<RW, 15>(c1);
    El c2, v2;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c2.v[t] = (u32)(i32)(v1.w[t] >> 29);
    const El cin2 = R_::template up<1>(c2);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v2.v[t] = ((u32)v1.w[t] & M) + cin2.v[t];
    col16 += (i64)R_::template ln<RW, 15>(c2);
    const i32 p16 = (i32)((u32)col16 & M);
    const i32 p17 = (i32)(col16 >> 29);
    // digits 9 .. 17 fold onto 0 .. 8: nine products below 2^59 per lane, two accumulators
    W64 f0, f1;
    const El k0 = c_fold<0>(), k1 = c_fold<1>(), k2 = c_fold<2>(), k3 = c_fold<3>(), k4 = c_fold<4>(),
             k5 = c_fold<5>(), k6 = c_fold<6>(), k7 = c_fold<7>(), k8 = c_fold<8>();
    const i32 h0 = R_::template ln<RW, 9>(v2), h1 = R_::template ln<RW, 10>(v2), h2 = R_::template ln<RW, 11>(v2),
              h3 = R_::template ln<RW, 12>(v2), h4 = R_::template ln<RW, 13>(v2), h5 = R_::template ln<RW, 14>(v2),
              h6 = R_::template ln<RW, 15>(v2);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      i64 x = (i64)s(v2.v[t] & live.v[t]);
      x += (i64)h0 * (i64)s(k0.v[t]);
      x += (i64)h2 * (i64)s(k2.v[t]);
      x += (i64)h4 * (i64)s(k4.v[t]);
      x += (i64)h6 * (i64)s(k6.v[t]);
      x += (i64)p17 * (i64)s(k8.v[t]);
      i64 y = (i64)h1 * (i64)s(k1.v[t]);
      y += (i64)h3 * (i64)s(k3.v[t]);
      y += (i64)h5 * (i64)s(k5.v[t]);
      y += (i64)p16 * (i64)s(k7.v[t]);
      f0.w[t] = x;
      f1.w[t] = y;
    }
    W64 f;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) f.w[t] = f0.w[t] + f1.w[t];
    // exact digits (the top digit keeps everything above it: below 2^58), then what stands above
    // 2^PBITS -- 34 bits at most -- times top = 2^PBITS mod p
    f = carry64(f);
    f = carry64(f);
    const i64 tv = R_::template ln64<RW, 8>(f);
    const i64 tt = tv >> TB;
    const i32 t0 = (i32)((u32)tt & M), t1 = (i32)(tt >> 29);
    const El ct = c_top(), ct1 = c_top1();
    W64 g;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      i64 x = f.w[t];
      if (lane_of(t) == TL) x = (i64)((u64)x & (((u64)1 << TB) - 1));
      x += (i64)t0 * (i64)s(ct.v[t]);
      x += (i64)t1 * (i64)s(ct1.v[t]);
      g.w[t] = x;
    }
    g = carry64(g);
    g = carry64(g);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (u32)g.w[t] & live.v[t];
    return r;
  }
#if defined(ELL_BOUNDS_CHECK)
  static void check(const El& a, const El& b, const char* what) {
    __int128 col[17] = {0};
    FpK256L::check_operands(R_::gather(a), R_::gather(b), col);
    FpK256L::check_cols(col);
    for (int t = 9; t < CL; t++) assert(a.v[t] == 0 && b.v[t] == 0 && "fpfoldc: dead lane not zero");
    (void)what;
  }
#endif
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    check(a, b, "mul");
#endif
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::columns(acc, col16, a, b);
    return reduce(acc, col16);
  }
  ELL_HD static El sqr(const El& a) { return mul(a, a); }
  ELL_HD static Q mulq(const Q& a, const Q& b) {
    Q r;
#if defined(__HIP_DEVICE_COMPILE__)
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::template columns<true>(acc, col16, a.r[0], b.r[0]);
    r.r[0] = reduce<true>(acc, col16);
#else
    for (int j = 0; j < R_::QR; j++) r.r[j] = mul(a.r[j], b.r[j]);
#endif
    return r;
  }
  // ---- conversions: plain residues, no Montgomery factor ---------------------------------------
  ELL_HD static void to_plain(u32 (&out)[L], const El& a) {
    u32 w[8];
    B::canon(w, a);
    ELL_UNROLL
    for (int j = 0; j < L; j++) out[j] = w[j];
  }
  // (an ABI value may exceed p: a product with 1 reduces it)
  ELL_HD static El from_plain(const u32 (&a)[L]) {
    u32 w[8];
    ELL_UNROLL
    for (int j = 0; j < 8; j++) w[j] = j < L ? a[j] : 0u;
    return mul(R_::scatter(FpK256L::from_plain(w)), one());
  }
  // an entry of the one-lane kernels' tables (canonical words) -> the digits, as they are
  ELL_HD static El load_words(const u32* w) {
    return B::each([&](int l) {
      const int ll = l > 8 ? 8 : l;
      const int bit = 29 * ll;
      const int k = bit >> 5, sh = bit & 31;
      const u32 lo = k < L ? w[k] : 0u;
      const u32 hi = k + 1 < L ? w[k + 1] : 0u;
      const u64 two = (u64)lo | ((u64)hi << 32);
      return l > 8 ? 0 : (i32)((u32)(two >> sh) & M);
    });
  }
  static ELL_HD_NOINLINE El inv(const El& a) {
    u32 w[L];
    to_plain(w, a);
    typename F1::El y = F1::inv(F1::from_plain(w));
    u32 v[L];
    F1::to_plain(v, y);
    return from_plain(v);
  }
};

}  // namespace ell
