// ellgpu -- the LANES-PER-ITEM layer for batches that leave the machine idle: one secp256k1
// field element spread over the lanes of ONE 16-lane DPP row, a whole wavefront per item.
//
// Why (profiles/r05_coop_field_microbench.json).  The reference's API is one
// item per call (curve/short.js:422-432, ec/index.js:188-229): a lone EC#verify or Point#mul is
// ONE dependent chain of field operations, and the device idles beside it.  An instruction costs
// a wave its ~5 SIMD cycles whether one lane is active or sixty-four, so what the single call
// waits for is INSTRUCTIONS PER FIELD OPERATION on its critical path: the one-item-per-lane
// product (fp.h / mul_asm.h) is 72 multiply-adds + ~100 carry and move instructions, an addition
// a 13-instruction carry chain.  Here limb l of an element (29 bits, signed: the radix and the
// lazy algebra of fpk256l.h) lives in lane l of the row:
//   * a product's nine partial-product rows are nine v_mad_i64_i32 of ALL lanes at once -- column
//     l accumulates in lane l; operand a's limbs arrive as SGPRs (v_readlane), operand b shifted
//     along the row by DPP row_shr -- column 16 (a8 b8) falls off the row onto the scalar unit;
//   * carries travel one lane up by DPP, twice, and the high columns fold back through
//     2^261 = 256 * 2^29 + 31264 (mod p) with three more multiply-adds and one carry pass;
//   * an addition or subtraction is ONE instruction, a normalisation eight.
// Measured on a lone wave (MI355X): product 184 ns against 439, square 180 against 362, Jacobian
// doubling 1.43 us against 3.21 -- 2.2-2.4x per operation.  Throughput per INSTRUCTION is
// the same, so this layer only serves batches of at most a few hundred items
// (Engine::Tuning::coop_grid); everything larger stays one item per lane.
//
// The group law and the ladders are the SAME templates the one-lane kernels instantiate
// (short.h dbl_lazy / add_mixed_lazy, ladder.h build_table_odd8 / run_odd_w4): FpK256C offers
// FpK256L's interface, with `El` holding this lane's limb.  All control flow is wave-uniform
// (one item per wave); zero tests reduce across the row through the scalar unit.
//
// Host passes (hipcc's host side, tests/hostsim) simulate the row: El holds all sixteen lanes
// and every primitive is a loop over them -- the CPU suite runs the same code, bounds checks
// (ELL_BOUNDS_CHECK) included.
//
// Replaces, for one item per wave: JPoint#dbl / mixedAdd (short.js:569-603, 668-737) and the
// bn.js Red / K256 arithmetic under them (dist/elliptic.js:6888-6931, 7078-7302).
#pragma once

#include "curves.h"

namespace ell {

#if defined(__HIP_DEVICE_COMPILE__)
#define ELL_COOP_LANES 1
#else
#define ELL_COOP_LANES 16
#endif

// PR = false: ONE item per wave -- the wave's four rows hold the same element (or, inside a step of
// the group law, four different elements of that item: Q / mulq); what is read out of a row is
// wave-uniform and travels through the scalar unit (v_readlane).
// PR = true ("one item per ROW", FpK256R): each of the wave's four 16-lane rows holds a DIFFERENT
// item, four items per wave.  Every product is the mulq instruction stream (operand limbs by DPP
// row_newbcast along the item's own row), nothing crosses a row, what is read out of a row is a
// vector register that differs from row to row, and control flow is row-coherent instead of
// wave-uniform.  The group law then runs its products one after the other (no Q: the four rows are
// taken), for about the instructions the PR = false form spends on ONE item (it pays for packing
// and unpacking its Qs) -- the layer for batches that four waves per SIMD of one-item waves no
// longer hold (Engine::Tuning::row_grid).
// The rows / lanes of a wave as a functor of the one-item-per-ROW layer walks them: on the device a
// lane IS one row / one lane (the loop body runs once, for this thread's); host passes walk all four
// rows / sixty-four lanes, one after the other.
#if defined(__HIP_DEVICE_COMPILE__)
#define ELL_FOR_ROWS(r) for (int r = (int)(threadIdx.x >> 4), r##_once_ = 1; r##_once_; r##_once_ = 0)
#define ELL_FOR_WAVE_LANES(l) for (int l = (int)(threadIdx.x & 63u), l##_once_ = 1; l##_once_; l##_once_ = 0)
#else
#define ELL_FOR_ROWS(r) for (int r = 0; r < 4; r++)
#define ELL_FOR_WAVE_LANES(l) for (int l = 0; l < 64; l++)
#endif

template <bool PR>
struct FpK256CT {
  static constexpr bool PER_ROW = PR;
  static constexpr int CL = ELL_COOP_LANES;        // lanes of the row held by one El: 1 = the hardware lane
  static constexpr int ROW = 16;
  static constexpr int L = 8;                      // 32-bit words of a plain value
  static constexpr bool LAZY = true;
  static constexpr bool HAS_SQRT = false;
  typedef Fe<CL> El;                               // v[t]: the two's complement bits of lane t's signed limb
  struct W64 { i64 w[CL]; };
  static constexpr u32 M = (1u << 29) - 1;
  static constexpr i32 R0 = 31264, R1 = 256;       // 2^261 = R1 * 2^29 + R0  (mod p)

  ELL_HD static i32 s(u32 x) { return (i32)x; }
  ELL_HD static void get_p(u32 (&p)[8]) { FpK256::get_p(p); }

  // ---- the row: lane index, per-lane constants, movement along it ---------------------------------
  ELL_HD static int lane_of(int t) {
#if defined(__HIP_DEVICE_COMPILE__)
    (void)t;
    return (int)(threadIdx.x & 15u);
#else
    return t;
#endif
  }
  // per-lane constant f(lane)
  template <class Fn>
  ELL_HD static El each(const Fn& f) {
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (u32)f(lane_of(t));
#if defined(__HIP_DEVICE_COMPILE__)
    // (an opaque register: without it the compiler folds a constant's lane cases into its users
    // and re-derives them, with branches, inside the ladders' loops)
    asm("" : "+v"(r.v[0]));
#endif
    return r;
  }
  // lane l <- lane l - N, zeros shifted in (DPP row_shr:N bound_ctrl:0)
  template <int N>
  ELL_HD static El up(const El& x) {
    El r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.v[0] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[0], 0x110 + N, 0xF, 0xF, true);
#else
    for (int t = 0; t < CL; t++) r.v[t] = t >= N ? x.v[t - N] : 0u;
#endif
    return r;
  }
  // lane l <- lane l + N, zeros shifted in (row_shl:N)
  template <int N>
  ELL_HD static El down(const El& x) {
    El r;
#if defined(__HIP_DEVICE_COMPILE__)
    r.v[0] = (u32)__builtin_amdgcn_update_dpp(0, (int)x.v[0], 0x100 + N, 0xF, 0xF, true);
#else
    for (int t = 0; t < CL; t++) r.v[t] = t + N < CL ? x.v[t + N] : 0u;
#endif
    return r;
  }
  // the (wave-uniform) value of lane l
  ELL_HD static i32 at(const El& x, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_readlane((int)x.v[0], l);
#else
    return s(x.v[l]);
#endif
  }
  // the value of lane I of THIS lane's row: wave-uniform through the scalar unit (PR = false), a
  // broadcast along the row (PR = true)
  template <int I>
  ELL_HD static i32 atc(const El& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (PR) return __builtin_amdgcn_update_dpp(0, (int)x.v[0], 0x150 + I, 0xF, 0xF, false);
    else return __builtin_amdgcn_readlane((int)x.v[0], I);
#else
    return s(x.v[I]);
#endif
  }
  // limb I as every lane of the row sees it.  RW = false: the element is the same in the four rows
  // of the wave (every value outside a `Q` is), lane I of row 0 arrives as a scalar register
  // (v_readlane); RW = true: each row holds an element of its own, lane I of THIS row is broadcast
  // along it (DPP row_newbcast) -- the value is a vector register, the arithmetic around it the same
  template <bool RW, int I>
  ELL_HD static i32 ln(const El& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (RW) return __builtin_amdgcn_update_dpp(0, (int)x.v[0], 0x150 + I, 0xF, 0xF, false);
    else return __builtin_amdgcn_readlane((int)x.v[0], I);
#else
    return s(x.v[I]);
#endif
  }
  // lane LANE <- sv (in every row of the wave: the rows stay copies of each other)
  template <int LANE>
  ELL_HD static El put(El x, i32 sv) {
#if defined(__HIP_DEVICE_COMPILE__)
    x.v[0] = lane_of(0) == LANE ? (u32)sv : x.v[0];
#else
    x.v[LANE] = (u32)sv;
#endif
    return x;
  }
  template <int N>
  ELL_HD static W64 up64(const W64& x) {
    W64 r;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)x.w[0], 0x110 + N, 0xF, 0xF, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((u64)x.w[0] >> 32), 0x110 + N, 0xF, 0xF, true);
    r.w[0] = (i64)(((u64)hi << 32) | lo);
#else
    for (int t = 0; t < CL; t++) r.w[t] = t >= N ? x.w[t - N] : 0;
#endif
    return r;
  }
  ELL_HD static i64 at64(const W64& x, int l) {
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)x.w[0], l);
    const u32 hi = (u32)__builtin_amdgcn_readlane((int)(u32)((u64)x.w[0] >> 32), l);
    return (i64)(((u64)hi << 32) | lo);
#else
    return x.w[l];
#endif
  }
  template <bool RW, int I>
  ELL_HD static i64 ln64(const W64& x) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (RW) {
      const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)x.w[0], 0x150 + I, 0xF, 0xF, false);
      const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)((u64)x.w[0] >> 32), 0x150 + I, 0xF, 0xF, false);
      return (i64)(((u64)hi << 32) | lo);
    } else {
      return at64(x, I);
    }
#else
    return x.w[I];
#endif
  }

  // ---- four elements side by side: one per ROW of the wavefront ----------------------------------------
  // A wave has four 16-lane rows and the item needs one.  Every element outside a Q is held by all
  // four rows alike (the same instruction computes it four times); a Q holds four DIFFERENT
  // elements, row j its own, so that ONE product instruction stream multiplies four independent
  // pairs (mulq) -- the group law below runs its independent products of a step that way
  // (short.h dbl_lazy / add_mixed_lazy).  On the device a Q is one register; host passes keep the
  // four rows as four elements.
  static constexpr bool QUAD = !PR;                // (one item per row: the rows are taken)
  static constexpr int QR = CL == 1 ? 1 : 4;
  struct Q { El r[QR]; };
  // rows (a, b, a, b)
  ELL_HD static Q pack2(const El& a, const El& b) {
    Q q;
#if defined(__HIP_DEVICE_COMPILE__)
    q.r[0].v[0] = (threadIdx.x & 16u) ? b.v[0] : a.v[0];
#else
    q.r[0] = a; q.r[1] = b; q.r[2] = a; q.r[3] = b;
#endif
    return q;
  }
  // rows (a, b, c, c)
  ELL_HD static Q pack3(const El& a, const El& b, const El& c) {
    Q q;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 ab = (threadIdx.x & 16u) ? b.v[0] : a.v[0];
    q.r[0].v[0] = (threadIdx.x & 32u) ? c.v[0] : ab;
#else
    q.r[0] = a; q.r[1] = b; q.r[2] = c; q.r[3] = c;
#endif
    return q;
  }
  // rows (a, b, c, d)
  ELL_HD static Q pack4(const El& a, const El& b, const El& c, const El& d) {
    Q q;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 ab = (threadIdx.x & 16u) ? b.v[0] : a.v[0];
    const u32 cd = (threadIdx.x & 16u) ? d.v[0] : c.v[0];
    q.r[0].v[0] = (threadIdx.x & 32u) ? cd : ab;
#else
    q.r[0] = a; q.r[1] = b; q.r[2] = c; q.r[3] = d;
#endif
    return q;
  }
  // every row <- rows 0 and 1 of a (.., .., same, same) Q: v_permlane16_swap trades the odd rows
  // of its first operand for the even rows of its second
  ELL_HD static void unpack2(const Q& q, El& a, El& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto p = __builtin_amdgcn_permlane16_swap(q.r[0].v[0], q.r[0].v[0], false, false);
    a.v[0] = p[0];
    b.v[0] = p[1];
#else
    a = q.r[0]; b = q.r[1];
#endif
  }
  // every row <- rows 0, 1, 2: v_permlane32_swap trades the upper half of its first operand for the
  // lower half of its second, then the row swap within each half
  ELL_HD static void unpack3(const Q& q, El& a, El& b, El& c) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto h = __builtin_amdgcn_permlane32_swap(q.r[0].v[0], q.r[0].v[0], false, false);   // (r0 r1 r0 r1), (r2 r3 r2 r3)
    const auto p = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
    a.v[0] = p[0];
    b.v[0] = p[1];
    c.v[0] = t[0];
#else
    a = q.r[0]; b = q.r[1]; c = q.r[2];
#endif
  }
  ELL_HD static void unpack4(const Q& q, El& a, El& b, El& c, El& d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const auto h = __builtin_amdgcn_permlane32_swap(q.r[0].v[0], q.r[0].v[0], false, false);
    const auto p = __builtin_amdgcn_permlane16_swap(h[0], h[0], false, false);
    const auto t = __builtin_amdgcn_permlane16_swap(h[1], h[1], false, false);
    a.v[0] = p[0];
    b.v[0] = p[1];
    c.v[0] = t[0];
    d.v[0] = t[1];
#else
    a = q.r[0]; b = q.r[1]; c = q.r[2]; d = q.r[3];
#endif
  }

  // per-lane constants of the algebra (loop-invariant registers on the device).  Written as masks,
  // not as ?: chains: a chain reaches the loop optimiser as control flow, which it does not hoist
  ELL_HD static u32 m_eq(int l, int i) { return 0u - (u32)(l == i); }
  ELL_HD static u32 m_lt(int l, int i) { return 0u - (u32)(l < i); }
  ELL_HD static El c_live() { return each([](int l) { return m_lt(l, 9); }); }
  ELL_HD static El c_mask() { return each([](int l) { return (m_lt(l, 8) & M) | m_eq(l, 8); }); }   // a carry pass keeps these bits
  // p - 2^256's part, limb by limb: K p = K * this (+ K 2^256 at limb 8)
  ELL_HD static El c_kp() { return each([](int l) { return (m_eq(l, 0) & (u32)-977) | (m_eq(l, 1) & (u32)-8) | (m_eq(l, 8) & (1u << 24)); }); }
  ELL_HD static El c_kf() { return each([](int l) { return (m_eq(l, 0) & 977u) | (m_eq(l, 1) & 8u); }); }       // one unit of limb 8's bit 24 (2^256 = 2^32 + 977)
  ELL_HD static El c_rr() { return each([](int l) { return (m_eq(l, 0) & (u32)R0) | (m_eq(l, 1) & (u32)R1); }); }        // one unit of column 9
  ELL_HD static El c_r1() { return each([](int l) { return m_lt(l, 9) & ~m_eq(l, 0) & (u32)R1; }); }
  ELL_HD static El c_rrr() { return each([](int l) { return (m_eq(l, 0) & (u32)(R1 * R0)) | (m_eq(l, 1) & (u32)(R1 * R1)); }); }   // one unit of column 17's R1 part (column 9 again)
  ELL_HD static El c_p() {
    return each([](int l) { return (m_eq(l, 0) & ((1u << 29) - 977u)) | (m_eq(l, 1) & ((1u << 29) - 9u)) | (m_lt(l, 8) & ~m_lt(l, 2) & M) | (m_eq(l, 8) & ((1u << 24) - 1u)); });
  }

  ELL_HD static El zero() { return each([](int) { return 0; }); }
  ELL_HD static El one() { return each([](int l) { return m_eq(l, 0) & 1u; }); }

  // ---- conversions ----------------------------------------------------------------------------------
  // the row <-> the one-lane 29-bit field (cold paths: canonical tests, stores)
  ELL_HD static FpK256L::El gather(const El& a) {
    FpK256L::El r;
    r.v[0] = (u32)atc<0>(a); r.v[1] = (u32)atc<1>(a); r.v[2] = (u32)atc<2>(a);
    r.v[3] = (u32)atc<3>(a); r.v[4] = (u32)atc<4>(a); r.v[5] = (u32)atc<5>(a);
    r.v[6] = (u32)atc<6>(a); r.v[7] = (u32)atc<7>(a); r.v[8] = (u32)atc<8>(a);
    return r;
  }
  ELL_HD static El scatter(const FpK256L::El& a) {
    return each([&](int l) {
      u32 v = 0;
      ELL_UNROLL
      for (int i = 0; i < 9; i++) v = l == i ? a.v[i] : v;
      return (i32)v;
    });
  }
  // wave-uniform plain words -> exact 29-bit digits (N form; the value may be >= p)
  ELL_HD static El from_plain(const u32 (&a)[8]) { return scatter(FpK256L::from_plain(a)); }
  // eight plain words IN MEMORY (an entry of the one-lane kernels' tables: fp.h FpK256 values,
  // 32-bit words little-endian) -> the row: lane l reads the two words its 29 bits straddle
  ELL_HD static El load_words(const u32* w) {
    return each([&](int l) {
      const int ll = l > 8 ? 8 : l;
      const int bit = 29 * ll, k = bit >> 5, sh = bit & 31;
      const u64 two = (u64)w[k] | ((u64)(k + 1 < 8 ? w[k + 1] : 0u) << 32);
      return l > 8 ? 0 : (i32)((u32)(two >> sh) & (l == 8 ? 0xFFFFFFu : M));
    });
  }
  // canonical residue in [0, p) as eight 32-bit words (wave-uniform)
  ELL_HD static void to_plain(u32 (&out)[8], const El& a) { FpK256L::to_plain(out, gather(a)); }
  ELL_HD static bool is_zero(const El& a) { return FpK256L::is_zero(gather(a)); }
  ELL_HD static bool eq(const El& a, const El& b) { return FpK256L::eq(gather(a), gather(b)); }
  ELL_HD static bool is_odd(const El& a) { return FpK256L::is_odd(gather(a)); }
  // Exact zero test for a DIRECT mul / sqr / mul2 output, whose value V lies in (-2^205, 2^256 +
  // 2^222): V = 0 (mod p) iff V is 0 or p, and V = limb 0 (mod 2^29) -- so unless limb 0's low 29
  // bits are those of 0 or of p (one output in 2^28), V is not zero; the canonical test decides
  // the rest.
  ELL_HD static bool is_zero_w(const El& a) {
    const u32 r0 = (u32)atc<0>(a) & M;
    if (ELL_UNLIKELY(r0 == 0u || r0 == M - 976u)) return is_zero(a);
    return false;
  }

  // ---- lazy primitives (fpk256l.h's, lane by lane) ------------------------------------------------------
  ELL_HD static El add_l(const El& a, const El& b) {
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = a.v[t] + b.v[t];
    return r;
  }
  // a - b + K p
  template <int K>
  ELL_HD static El sub_l(const El& a, const El& b) {
    const El kp = c_kp();
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = a.v[t] - b.v[t] + (u32)K * kp.v[t];
    return r;
  }
  template <int K>
  ELL_HD static El neg_l(const El& a) { return sub_l<K>(zero(), a); }
  // c ? K p - a : a
  template <int K>
  ELL_HD static El cneg_l(const El& a, bool c) {
    const El n = neg_l<K>(a);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = c ? n.v[t] : a.v[t];
    return r;
  }
  // parallel carry pass + top fold (fpk256l.h norm): any lazy value with |limbs| < 2^31 -> N form
  ELL_HD static El norm(const El& a) {
    const El mk = c_mask(), kf = c_kf(), live = c_live();
    El c;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c.v[t] = (u32)(s(a.v[t]) >> 29);
    const El cin = up<1>(c);
    i32 r8;
    const i32 f = FpK256L::top_fold(atc<8>(a), r8);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      u32 v = (a.v[t] & mk.v[t]) + cin.v[t] + (u32)f * kf.v[t];
      if (lane_of(t) == 8) v -= (u32)f << 24;
      r.v[t] = v & live.v[t];                          // (lane 9 received limb 8's carry: limb 8 keeps it)
    }
    return r;
  }
  // (a << K) in N form, a in N form (K <= 3)
  template <int K>
  ELL_HD static El shl_norm(const El& a) {
    const El mk = c_mask(), kf = c_kf(), live = c_live();
    El c;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c.v[t] = (u32)(s(a.v[t]) >> (29 - K));
    const El cin = up<1>(c);
    i32 r8;
    const i32 f = FpK256L::top_fold(atc<8>(a) << K, r8);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      u32 v = ((a.v[t] << K) & mk.v[t]) + cin.v[t] + (u32)f * kf.v[t];
      if (lane_of(t) == 8) v -= (u32)f << 24;
      r.v[t] = v & live.v[t];
    }
    return r;
  }
  // a / 2 mod p for a lazy value with |limbs| < 2^30 (fpk256l.h half_l)
  ELL_HD static El half_l(const El& a) {
    const El pv = c_p();
    const u32 odd = 0u - ((u32)atc<0>(a) & 1u);
    El tt;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) tt.v[t] = a.v[t] + (odd & pv.v[t]);
    const El nx = down<1>(tt);                         // (lane 8 sees lane 9: zero)
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (u32)(s(tt.v[t]) >> 1) + ((nx.v[t] & 1u) << 28);
    return r;
  }

  // ---- products -------------------------------------------------------------------------------------------
#if defined(ELL_BOUNDS_CHECK)
  static void check(const El& a, const El& b, __int128 (&col)[17], const char* what) {
    FpK256L::check_operands(gather(a), gather(b), col);
    for (int t = 9; t < CL; t++)
      if (a.v[t] != 0 || b.v[t] != 0) { fprintf(stderr, "fpk256c %s: dead lane %d is not zero\n", what, t); assert(0); }
  }
#endif
  // columns 0..15 of a * b onto acc (one per lane), column 16 onto col16
  // (RW: a and b differ from row to row -- see ln)
  template <bool RW = false>
  ELL_HD static void columns(W64& acc, i64& col16, const El& a, const El& b) {
    const i32 a0 = ln<RW, 0>(a), a1 = ln<RW, 1>(a), a2 = ln<RW, 2>(a), a3 = ln<RW, 3>(a), a4 = ln<RW, 4>(a),
              a5 = ln<RW, 5>(a), a6 = ln<RW, 6>(a), a7 = ln<RW, 7>(a), a8 = ln<RW, 8>(a);
    const El b1 = up<1>(b), b2 = up<2>(b), b3 = up<3>(b), b4 = up<4>(b), b5 = up<5>(b), b6 = up<6>(b),
             b7 = up<7>(b), b8 = up<8>(b);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      i64 c = acc.w[t];
      c += (i64)a0 * s(b.v[t]);
      c += (i64)a1 * s(b1.v[t]);
      c += (i64)a2 * s(b2.v[t]);
      c += (i64)a3 * s(b3.v[t]);
      c += (i64)a4 * s(b4.v[t]);
      c += (i64)a5 * s(b5.v[t]);
      c += (i64)a6 * s(b6.v[t]);
      c += (i64)a7 * s(b7.v[t]);
      c += (i64)a8 * s(b8.v[t]);
      acc.w[t] = c;
    }
    col16 += (i64)a8 * (i64)ln<RW, 8>(b);
  }
  // carries and the fold of columns 9.. -> N form
  // (bounds, for column sums below 2^63: pass 1 leaves limbs below 2^29 + 2^35, pass 2 below 2^29 +
  // 2^6; column 16 splits into a 29-bit digit and a part below 2^24; the folded limbs stay below
  // 2^47, their carries below 2^18; limb 0 ends below 2^29 + 2^25, the others below 2^29 + 2^19,
  // limb 8 in [0, 2^24) -- see DESIGN.md section 4)
  template <bool RW = false>
  ELL_HD static El tail(const W64& acc, i64 col16) {
    const El r1 = c_r1(), rr = c_rr(), rrr = c_rrr(), kf = c_kf(), live = c_live();
    // first carry pass, 64-bit carries
    W64 c1;
    El lo1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) { c1.w[t] = acc.w[t] >> 29; lo1.v[t] = (u32)acc.w[t] & M; }
    const W64 cin1 = up64<1>(c1);
    W64 v1;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v1.w[t] = (i64)lo1.v[t] + cin1.w[t];
    col16 += ln64<RW, 15>(c1);
    // second pass: the carries fit a word
    El c2, v2;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c2.v[t] = (u32)(i32)(v1.w[t] >> 29);
    const El cin2 = up<1>(c2);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v2.v[t] = ((u32)v1.w[t] & M) + cin2.v[t];
    col16 += (i64)ln<RW, 15>(c2);
    const i32 p16 = (i32)((u32)col16 & M);
    const i32 p17 = (i32)(col16 >> 29);
    // fold: column 9 + j -> R0 at limb j, R1 at limb j + 1; column 17's R1 part is column 9 again
    El h0 = down<9>(v2);                               // lane j <- column 9 + j   (j <= 6)
    h0 = put<7>(h0, p16);
    h0 = put<8>(h0, p17);
    El h1 = down<8>(v2);                               // lane j <- column 8 + j   (lane 0: times 0 below)
    h1 = put<8>(h1, p16);
    W64 tt;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      i64 x = (i64)s(v2.v[t]);
      x += (i64)s(h0.v[t]) * R0;
      x += (i64)s(h1.v[t]) * s(r1.v[t]);
      x += (i64)p17 * s(rrr.v[t]);
      tt.w[t] = x;
    }
    // carry pass over the folded limbs
    El c3, v3;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) c3.v[t] = (u32)(i32)(tt.w[t] >> 29);
    const El cin3 = up<1>(c3);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v3.v[t] = ((u32)tt.w[t] & M) + cin3.v[t];
    // limb 8: its carry is column 9 once more (small now), its bits above 2^24 fold through 2^256
    const i32 c9 = ln<RW, 8>(c3);
    const i32 x8 = ln<RW, 8>(v3);
    const i32 hi = x8 >> 24;
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      u32 v = v3.v[t] + (u32)c9 * rr.v[t] + (u32)hi * kf.v[t];
      if (lane_of(t) == 8) v -= (u32)hi << 24;
      r.v[t] = v & live.v[t];
    }
    return r;
  }
  ELL_HD static W64 zero64() {
    W64 z;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) z.w[t] = 0;
    return z;
  }
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    { __int128 col[17] = {0}; check(a, b, col, "mul"); FpK256L::check_cols(col); }
#endif
    W64 acc = zero64();
    i64 col16 = 0;
    columns<PR>(acc, col16, a, b);
    return tail<PR>(acc, col16);
  }
  ELL_HD static El sqr(const El& a) { return mul(a, a); }
  // row j of the result = row j of a * row j of b: four products for the instructions of one
  ELL_HD static Q mulq(const Q& a, const Q& b) {
    Q r;
#if defined(__HIP_DEVICE_COMPILE__)
    W64 acc = zero64();
    i64 col16 = 0;
    columns<true>(acc, col16, a.r[0], b.r[0]);
    r.r[0] = tail<true>(acc, col16);
#else
    for (int j = 0; j < QR; j++) r.r[j] = mul(a.r[j], b.r[j]);
#endif
    return r;
  }
  // a * b + e * f with one reduction
  ELL_HD static El mul2(const El& a, const El& b, const El& e, const El& f) {
#if defined(ELL_BOUNDS_CHECK)
    { __int128 col[17] = {0}; check(a, b, col, "mul2"); check(e, f, col, "mul2"); FpK256L::check_cols(col); }
#endif
    W64 acc = zero64();
    i64 col16 = 0;
    columns<PR>(acc, col16, a, b);
    columns<PR>(acc, col16, e, f);
    return tail<PR>(acc, col16);
  }

  // a^-1 (0 for 0): canonical words -> the division steps of the saturated field -> back (cold: once
  // per item, where a result leaves the row layer in affine form)
  static ELL_HD_NOINLINE El inv(const El& a) {
    FpK256::El t;
    to_plain(t.v, a);
    FpK256::El r = FpK256::inv(t);
    return from_plain(r.v);
  }

  // ---- generic interface: N form in, N form out --------------------------------------------------------
  ELL_HD static El add(const El& a, const El& b) { return norm(add_l(a, b)); }
  ELL_HD static El sub(const El& a, const El& b) { return norm(sub_l<4>(a, b)); }
  ELL_HD static El neg(const El& a) { return norm(neg_l<4>(a)); }
  ELL_HD static El dbl(const El& a) { return shl_norm<1>(a); }
  template <int K>
  ELL_HD static El mul_pow2(const El& a) { return shl_norm<K>(a); }
};

typedef FpK256CT<false> FpK256C;
typedef FpK256CT<true> FpK256R;

// the curve as the group-law templates see it (short.h ShortOps, ladder.h Ladder)
template <bool PR>
struct CvSecp256k1CT {
  typedef FpK256CT<PR> F;
  typedef FpMont<consts::SECP256K1_N> Fn;
  typedef consts::SECP256K1_C C;
  static constexpr int A_KIND = 0;
  static constexpr bool ENDO = true;
  static constexpr bool JTABLE = false;
  static constexpr int ID = CURVE_SECP256K1;
};
typedef CvSecp256k1CT<false> CvSecp256k1C;
typedef CvSecp256k1CT<true> CvSecp256k1R;

}  // namespace ell
