// ellgpu -- the lanes-per-item layer for the WIDE NIST primes (p384, p521): one field element over
// the lanes of a whole WAVE -- digit l (28 bits, signed) in lane l -- where coop.h / coop_mont.h
// keep an element inside one 16-lane DPP row (nine 29-bit digits: at most 261 bits).
//
// Why.  The reference's API is one item per call (curve/short.js:422-432, curve/base.js:86-126,
// ec/index.js:188-229); a lone p384 / p521 EC#verify or Point#mul is ONE dependent chain of field
// operations, and on the one-item-per-lane kernels every product of that chain is 144 / 289
// multiply-adds plus a Solinas / shift-add fold of a few hundred instructions (4.7 / 9.5 ms per
// verify through install(), profiles/r05_js_single_call.jsonl -- 3.0-3.8x the JavaScript's own
// time only because bn.js has no fast reduction for these primes).  A 14- or 19-digit element does
// not fit a 16-lane row (27 / 37 product columns), but it fits the wave's 64 lanes:
//   * product: column c accumulates in lane c -- NL multiply-adds (v_mad_i64_i32) of the whole wave,
//     operand a's digits as scalars (v_readlane), operand b moving up ONE lane per step (DPP
//     wave_shr:1, which crosses the row boundaries a row_shr does not);
//   * reduction by FOLDING, no serial chain: after two carry passes digit NL + j of the product is
//     a 28-bit number worth fold[j] = 2^(28 (NL + j)) mod p, a per-lane constant vector -- NL more
//     multiply-adds of the whole wave (the digit as a scalar: v_readlane); what then stands above
//     2^PBITS (34 bits at most) is worth top = 2^PBITS mod p (2^128 + 2^96 - 2^32 + 1, or 1) and
//     folds once more.  (The first version of this layer reduced word-serially as coop_mont.h does,
//     Montgomery form with R = 2^(28 NL): 14 / 19 steps that each wait for v_readlane -> scalar ->
//     v_mad_i64_i32 of the step before -- 240 / 320 instructions at 7 cycles each on a lone wave,
//     p384 EC#verify 2.69 ms, p521 4.77; profiles/r06_wide_field.jsonl.)
//   * an addition or subtraction is one instruction + a seven-instruction normalisation.
// 28-bit digits (not 29): a column sums up to 19 products -- 2^61.3.  Plain residues (no Montgomery
// form), signed digits, values in (-2^(PBITS-20), 2^PBITS + 2^(PBITS-20)), nothing conditionally
// subtracted on the fast path.  The interface is coop_mont.h FpMontC's, so coop_work.h CoopNist,
// short.h and ladder.h instantiate over it unchanged; QUAD is off (one element per wave).
//
// Host passes (tests/hostsim) simulate the wave: El holds all 64 lanes.
//
// Replaces, for one item per wave: JPoint#dbl / mixedAdd / add (short.js:532-603, 739-800) over
// bn.js `Mont` with 15- / 21-word numbers (dist/elliptic.js:7308-7381, 4941-5560 smallMulTo).
#pragma once

#include "coop.h"
#include "coop_consts.h"

namespace ell {

#if defined(__HIP_DEVICE_COMPILE__)
#define ELL_WIDE_LANES 1
#else
#define ELL_WIDE_LANES 64
#endif

template <class MC, class F1>
struct FpFoldW {
  static constexpr int CL = ELL_WIDE_LANES;
  static constexpr int SPAN = 64;                    // lanes an element may touch: the wave
  static constexpr int NL = MC::NL;
  static constexpr int RB = MC::RB;
  static constexpr int L = F1::L;                    // 32-bit words of a plain value
  static constexpr bool HAS_SQRT = false;
  static constexpr bool WIDE = true;
  typedef Fe<CL> El;
  struct W64 { i64 w[CL]; };
  static constexpr u32 M = (1u << RB) - 1;
  static constexpr int TL = NL - 1;                  // the digit that holds bit PBITS - 1 ...
  static constexpr int TB = MC::PBITS - RB * TL;     // ... and how many bits of it belong to a value below 2^PBITS
  static_assert((MC::PBITS - 1) / RB == TL && 2 * NL <= SPAN, "FpFoldW: top digit NL - 1, columns within the wave");

  ELL_HD static i32 s(u32 x) { return (i32)x; }

  // ---- the wave as a row -------------------------------------------------------------------------
  ELL_HD static int lane_of(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)t;
    return (int)(threadIdx.x & 63u);
#else
    return t;
#endif
  }
  template <class Fn>
  ELL_HD static El each(const Fn& f) {
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (u32)f(lane_of(t));
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(r.v[0]));                            // (an opaque register: see coop.h each)
#endif
    return r;
  }
  // lane l <- lane l - 1 across the whole wave, zero into lane 0 (DPP wave_shr:1, bound_ctrl:0)
  ELL_HD static El up1(const El& x) {
    El r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.v[0] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[0], 0x138, 0xF, 0xF, true);
#else
    for (int t = 0; t < CL; t++) r.v[t] = t >= 1 ? x.v[t - 1] : 0u;
#endif
    return r;
  }
  ELL_HD static W64 up1_64(const W64& x) {
    W64 r;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)x.w[0], 0x138, 0xF, 0xF, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((u64)x.w[0] >> 32), 0x138, 0xF, 0xF, true);
    r.w[0] = (i64)(((u64)hi << 32) | lo);
#else
    for (int t = 0; t < CL; t++) r.w[t] = t >= 1 ? x.w[t - 1] : 0;
#endif
    return r;
  }
  // the (wave-uniform) value of lane I
  template <int I>
  ELL_HD static i32 at(const El& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane((int)x.v[0], I);
#else
    return s(x.v[I]);
#endif
  }
  template <int I>
  ELL_HD static i64 at64(const W64& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x.w[0], I);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)x.w[0] >> 32), I);
    return (i64)(((u64)hi << 32) | lo);
#else
    return x.w[I];
#endif
  }

  // ---- per-lane constants (loop-invariant registers on the device; masks, not ?: chains) ----------
  ELL_HD static u32 m_eq(int l, int i) { return 0u - (u32)(l == i); }
  ELL_HD static u32 m_lt(int l, int i) { return 0u - (u32)(l < i); }
  // digits of a compile-time table, moved up I lanes
  template <int I, class Tab>
  ELL_HD static El by_lane(const Tab& tab) {
    return each([&](int l) {
      u32 v = 0;
      ELL_UNROLL
      for (int j = 0; j < NL; j++) v |= m_eq(l, j + I) & (u32)tab[j];
      return (i32)v;
    });
  }
  ELL_HD static El c_p() { return by_lane<0>(MC::pd); }
  template <int J>
  ELL_HD static El c_fold() { return by_lane<0>(MC::fold[J]); }   // what digit NL + J of a product is worth
  ELL_HD static El c_top() { return by_lane<0>(MC::top); }        // 2^PBITS mod p ...
  ELL_HD static El c_top1() { return by_lane<1>(MC::top); }       // ... and 2^(PBITS + RB) mod p (top is short)
  ELL_HD static El c_live() { return each([](int l) { return m_lt(l, NL); }); }
  ELL_HD static El c_mask() { return each([](int l) { return (m_lt(l, NL - 1) & M) | m_eq(l, NL - 1); }); }   // a carry pass keeps these bits
  ELL_HD static El c_carries() { return each([](int l) { return m_lt(l, NL - 1); }); }                          // digits that hand a carry up
  ELL_HD static El zero() { return each([](int) { return 0; }); }
  ELL_HD static El one() { return each([](int l) { return m_eq(l, 0) & 1u; }); }

  // ---- lazy digit-wise forms ----------------------------------------------------------------------
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
  // Carry pass + value fold: a lazy value with |digits| < 2^30 and |value| < 3 * 2^PBITS comes back
  // with digits in (-2^3, 2^28 + 2^3) and its value in (-2^(PBITS-20), 2^PBITS + 2^(PBITS-20)).
  // The fold subtracts k p for k = floor(value / 2^PBITS) as the top digit shows it, k in [-2, 2].
  ELL_HD static El norm(const El& a) {
    const El pv = c_p(), mk = c_mask(), live = c_live(), cm = c_carries();
    const i32 top = at<TL>(a) >> TB;
    const i32 k = top < -2 ? -2 : (top > 2 ? 2 : top);
#if defined(ELL_BOUNDS_CHECK)
    for (int t = 0; t < CL; t++) {
      const i64 dd = (i64)s(a.v[t]) - (i64)k * (i64)s(pv.v[t]);
      if (dd >= ((i64)1 << 31) || dd < -((i64)1 << 31)) { fprintf(stderr, "fpfoldw norm: digit %d leaves 32 bits\n", t); assert(0); }
    }
#endif
    El d, c;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) d.v[t] = a.v[t] - (u32)k * pv.v[t];
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c.v[t] = (u32)(s(d.v[t]) >> RB) & cm.v[t];          // (the top digit keeps its carry)
    const El cin = up1(c);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = ((d.v[t] & mk.v[t]) + cin.v[t]) & live.v[t];
    return r;
  }

  // ---- products -------------------------------------------------------------------------------------
#if defined(ELL_BOUNDS_CHECK)
  static void check(const El& a, const El& b, const char* what) {
    const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
    for (int c = 0; c < 2 * NL - 1; c++) {
      __int128 sum = 0;
      for (int i = 0; i < NL; i++) {
        const int j = c - i;
        if (j < 0 || j >= NL) continue;
        __int128 pr = (__int128)s(a.v[i]) * (__int128)s(b.v[j]);
        sum += pr < 0 ? -pr : pr;
      }
      if (sum >= lim) { fprintf(stderr, "fpfoldw %s: column %d exceeds 63 bits\n", what, c); assert(0); }
    }
    for (int t = NL; t < CL; t++) assert(a.v[t] == 0 && b.v[t] == 0 && "fpfoldw: dead lane not zero");
  }
  static void check64(const W64& x, const char* what) {
    const i64 lim = ((i64)1 << 62);
    for (int t = 0; t < CL; t++)
      if (x.w[t] >= lim || x.w[t] <= -lim) { fprintf(stderr, "fpfoldw %s: lane %d exceeds 62 bits\n", what, t); assert(0); }
  }
#endif
  // columns 0 .. 2 NL - 2 of a * b, one per lane: step I adds a_I * (b moved up I lanes); two
  // accumulators (even / odd steps), so that a step does not wait for the one before it
  template <int I>
  ELL_HD static void col_steps(W64& acc0, W64& acc1, const El& a, El bs) {
    if constexpr (I < NL) {
      const i32 ai = at<I>(a);
      W64& acc = (I & 1) ? acc1 : acc0;
      ELL_UNROLL
      for (int t = 0; t < CL; t++) acc.w[t] += (i64)ai * (i64)s(bs.v[t]);
      if constexpr (I + 1 < NL) col_steps<I + 1>(acc0, acc1, a, up1(bs));
    }
  }
  // + h_J * fold[J], J = 0 .. NL - 1: h_J = digit NL + J of the carried product
  template <int J>
  ELL_HD static void fold_steps(W64& acc0, W64& acc1, const El& d) {
    if constexpr (J < NL) {
      const El cj = c_fold<J>();                       // a per-lane CONSTANT (hoisted out of the ladders' loops)
      const i32 h = at<NL + J>(d);
      W64& acc = (J & 1) ? acc1 : acc0;
      ELL_UNROLL
      for (int t = 0; t < CL; t++) acc.w[t] += (i64)h * (i64)s(cj.v[t]);
      fold_steps<J + 1>(acc0, acc1, d);
    }
  }
  // one carry pass over 64-bit lanes: low RB bits stay, the rest moves one lane up.  KEEP: the
  // lanes whose value stays whole (the top of the range: it takes the carries and hands none on)
  ELL_HD static W64 carry64(const W64& x, const El& keep) {
    W64 c, lo;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      const bool k = keep.v[t] != 0;
      c.w[t] = k ? 0 : (x.w[t] >> RB);
      lo.w[t] = k ? x.w[t] : (i64)((u32)x.w[t] & M);
    }
    const W64 cin = up1_64(c);
    W64 r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.w[t] = lo.w[t] + cin.w[t];
    return r;
  }
  ELL_HD static W64 zero64() {
    W64 z;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) z.w[t] = 0;
    return z;
  }
  // Column sums below 2^62 in lanes 0 .. 2 NL - 2 -> the product's residue, digits in [0, 2^28 + 2^5),
  // value in [0, 2^PBITS + 2^(PBITS - 200)).
  ELL_HD static El reduce(const W64& cols) {
    const El live = c_live();
    const El top2 = each([](int l) { return m_eq(l, 2 * NL - 1); });     // column 2 NL - 1: carries only
    const El topn = each([](int l) { return m_eq(l, NL - 1); });
    // 1. the columns as digits: two carry passes (64-bit sums -> below 2^28 + 2^34 -> below 2^28 + 2^7);
    //    lane 2 NL - 1 collects what leaves column 2 NL - 2 (below 2^29)
    W64 v = carry64(cols, top2);
    v = carry64(v, top2);
    El d;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) d.v[t] = (u32)v.w[t];
    // 2. digits NL .. 2 NL - 1 fold onto 0 .. NL - 1: each is worth a constant vector (digits below 2^28):
    //    NL products below 2^58 per lane
    W64 f0, f1 = zero64();
    ELL_UNROLL
    for (int t = 0; t < CL; t++) f0.w[t] = (i64)s(d.v[t] & live.v[t]);
    fold_steps<0>(f0, f1, d);
    W64 f;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) f.w[t] = f0.w[t] + f1.w[t];
#if defined(ELL_BOUNDS_CHECK)
    check64(f, "fold");
#endif
    // 3. exact digits again (two passes; the top digit keeps everything above it: below 2^62), then
    //    what stands above 2^PBITS -- t, 35 bits at most -- is worth t * top: t = t1 2^28 + t0
    f = carry64(f, topn);
    f = carry64(f, topn);
    const i64 tv = at64<TL>(f);
    const i64 tt = tv >> TB;
    const i32 t0 = (i32)((u32)tt & M), t1 = (i32)(tt >> RB);
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
#if defined(ELL_BOUNDS_CHECK)
    check64(g, "top fold");
#endif
    // 4. digits once more: below 2^57 -> two passes
    g = carry64(g, topn);
    g = carry64(g, topn);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (u32)g.w[t] & live.v[t];
    return r;
  }
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    check(a, b, "mul");
#endif
    W64 acc0 = zero64(), acc1 = zero64();
    col_steps<0>(acc0, acc1, a, b);
    W64 acc;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) acc.w[t] = acc0.w[t] + acc1.w[t];
    return reduce(acc);
  }
  ELL_HD static El sqr(const El& a) { return mul(a, a); }

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
  template <int I>
  ELL_HD static void gather_steps(i64 (&d)[NL], const El& a) {
    if constexpr (I < NL) { d[I] = (i64)at<I>(a); gather_steps<I + 1>(d, a); }
  }
  // the value's canonical residue in [0, p) as exact digits -> L plain words
  ELL_HD static void to_plain(u32 (&out)[L], const El& a) {
    i64 d[NL];
    gather_steps<0>(d, a);
    // value in (-p, 2p): + p, sequential carry, then up to two conditional subtractions of p
    i64 c = 0;
    u32 dig[NL];
    ELL_UNROLL
    for (int i = 0; i < NL; i++) {
      i64 t = d[i] + (i64)MC::pd[i] + c;
      dig[i] = (u32)t & M;
      c = t >> RB;
    }
    ELL_NOUNROLL
    for (int it = 0; it < 2; it++) {
      u32 sub[NL];
      i32 br = 0;
      ELL_UNROLL
      for (int i = 0; i < NL; i++) {
        i32 t = (i32)dig[i] - MC::pd[i] + br;
        sub[i] = (u32)t & M;
        br = t >> RB;
      }
      const bool take = br == 0;                       // dig >= p
      ELL_UNROLL
      for (int i = 0; i < NL; i++) dig[i] = take ? sub[i] : dig[i];
    }
    u32 w[L + 2];
    ELL_UNROLL
    for (int j = 0; j < L + 2; j++) w[j] = 0;
    ELL_UNROLL
    for (int i = 0; i < NL; i++) {
      const int bit = RB * i, k = bit >> 5, sh = bit & 31;
      const u64 v = (u64)dig[i] << sh;
      if (k < L + 2) w[k] |= (u32)v;
      if (k + 1 < L + 2) w[k + 1] |= (u32)(v >> 32);
    }
    ELL_UNROLL
    for (int j = 0; j < L; j++) out[j] = w[j];
  }
  // wave-uniform plain words (any L-word value: it may exceed p) -> the element's digits
  ELL_HD static El from_plain(const u32 (&a)[L]) {
    u32 dig[NL];
    ELL_UNROLL
    for (int i = 0; i < NL; i++) {
      const int bit = RB * i, k = bit >> 5, sh = bit & 31;
      const u64 two = (u64)(k < L ? a[k] : 0u) | ((u64)(k + 1 < L ? a[k + 1] : 0u) << 32);
      dig[i] = (u32)(two >> sh) & (i == NL - 1 ? 0xFFFFFFFFu : M);     // (the top digit takes what is left of the words)
    }
    const El digs = each([&](int l) {
      u32 v = 0;
      ELL_UNROLL
      for (int i = 0; i < NL; i++) v |= m_eq(l, i) & dig[i];
      return (i32)v;
    });
    // (an ABI value may exceed p by a factor of 2^7 on p521 -- 66 bytes -- : a product with 1 reduces it)
    return mul(digs, one());
  }
  // L plain words in memory (an entry of the one-lane kernels' tables: canonical) -> the element: lane l
  // reads the two words its digit straddles
  ELL_HD static El load_words(const u32* w) {
    return each([&](int l) {
      const int ll = l >= NL ? NL - 1 : l;
      const int bit = RB * ll;
      const int k = bit >> 5, sh = bit & 31;
      const u32 lo = k < L ? w[k] : 0u;
      const u32 hi = k + 1 < L ? w[k + 1] : 0u;
      const u64 two = (u64)lo | ((u64)hi << 32);
      return l >= NL ? 0 : (i32)((u32)(two >> sh) & M);
    });
  }
  // Zero test.  Every value of the interface lies in (-p, 2p): it is 0 (mod p) iff it is 0 or p, and
  // then digit 0's low bits are 0's or p's -- anything else is not zero; the canonical digits decide
  // the rest.
  ELL_HD static bool is_zero(const El& a) {
    const u32 r0 = (u32)at<0>(a) & M;
    if (ELL_UNLIKELY(r0 == 0u || r0 == ((u32)MC::pd[0] & M))) {
      u32 w[L];
      to_plain(w, a);
      return bn_is_zero<L>(w);
    }
    return false;
  }
  ELL_HD static bool eq(const El& a, const El& b) { return is_zero(sub(a, b)); }
  // a^-1 (0 for 0): through the one-lane field's inversion
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
