// ellgpu -- prime-field arithmetic, one field element per lane, limbs in VGPRs.
//
// Field families, all exposing the same static interface (L, El, add, sub, neg, dbl,
// mul_pow2, mul, sqr, inv, is_zero, eq, from_plain, to_plain, one, zero):
//
//   FpK256      p = 2^256 - 2^32 - 977 (secp256k1): plain residues, the
//               512-bit product is folded with 2^256 == 2^32 + 977.
//   Fp25519     p = 2^255 - 19: plain residues, folded with 2^256 == 38.
//   FpSolinas   p256 / p384: plain residues, FIPS 186-4 D.2 word sums.
//   FpP521      p = 2^521 - 1: plain residues, Mersenne fold.
//   FpMont<P>   any odd modulus (p192, p224, and every group order n):
//               Montgomery residues (x*R mod p, R = 2^(32L)); wide product + row-wise REDC
//               on the device, CIOS in the host test build.
//
// They replace bn.js `Red`+`K256` / `Red`+`P25519` / `Mont` contexts
// (reference: dist/elliptic.js:6888-7381) -- internal representation is ours;
// every value that leaves a kernel is converted to the canonical residue, which
// is all the reference's results depend on.
//
// Every element is kept fully reduced in [0, p) after every operation, so
// is_zero/eq are limb compares and the exceptional branches of the group law
// can be detected exactly.
#pragma once

#include "common.h"
#include "curve_consts.h"
#include "mul_asm.h"
#include "safegcd.h"

// field multiplies of moduli with at least this many limbs are real function calls: only the
// 17-limb order field of p521 today.  The 12-limb fields of p384 were calls too (24 argument
// words fit the 32 argument VGPRs) until the p521 lesson was applied to them: inlined, and held
// to two waves per SIMD by the kernels' launch bounds, p384 gains 13-21 % (P*k 18.1 -> 20.5 M/s,
// G*k 114 -> 138 M/s, same box).
#ifndef ELL_MONT_CALL_MINL
#define ELL_MONT_CALL_MINL 13
#endif

namespace ell {

template <int N>
struct Fe {
  u32 v[N];
};

// Wide product / square of L-limb integers.  Device code uses the generated
// v_mad_u64_u32 + carry-out blocks of mul_asm.h; host passes (hipcc's host side
// and the CPU unit-test build) use the portable operand-scanning code.
// ELL_MUL_CHAIN = 1: single-accumulator-chain blocks (one column per asm statement, the
// compiler assembles the next column's (X.hi, E) pair); 0: two-column blocks + carry combines.
// ELL_P521_INLINE = 1: the 17-limb multiply / square are inlined like the smaller fields'.  As
// function calls their 34 + 17 operand words do not fit the 32 argument VGPRs, and everything
// live across a call goes through the stack: 1.2-1.6 KB of scratch per lane and half the
// throughput (3.6 M P*k/s against 6.8 M/s inlined, scratch 0).
#ifndef ELL_P521_INLINE
#define ELL_P521_INLINE 1
#endif
// ELL_P521_JTABLE = 1 switches p521 back to the signed-window ladder over a Jacobian table.  The
// odd-digit affine-table ladder of the smaller curves is faster there too (7.2 against 6.75 M
// P*k/s, same box) and has the smaller loop body -- the p521 kernels are sensitive to that: their
// inlined code no longer fits the instruction cache, and boxes of the pool whose memory side
// runs slower (normalize -20 %) lose 40 % on them instead of the usual 5 %.
#ifndef ELL_P521_JTABLE
#define ELL_P521_JTABLE 0
#endif
#ifndef ELL_MUL_CHAIN
#define ELL_MUL_CHAIN 1
#endif
// ELL_MUL_FAST = 1: the wide products first run the generated "fast" chains, in which the
// first multiply-add of every column does not feed its carry-out into the column's extension
// word: its addend (X.hi, E) of the previous column is below 2^36, so it overflows 64 bits only
// when a_i * b_j >= 2^64 - 2^36 -- never for random operands, always possible for chosen ones
// (limbs of 2^32 - 1).  The carry-out masks are OR-ed on the scalar unit instead, and a wave in
// which any lane did overflow redoes the product with the all-carries chain (a uniform branch,
// taken with probability ~2^-21 per product).  15 of 64 (8 limbs) carry additions less -- and
// MEASURED SLOWER (round 2, same box: field mul 149.5 -> 153.5 issue units at 3 waves/SIMD,
// ecdsa_main 8.57 -> 8.77 ms): the scalar ORs and the per-product scalar branch cost the wave
// more issue turns than the 15 v_addc they replace (an instruction of ANY kind costs a wave of
// this kernel ~4.3 cycles of SIMD time).  Kept as a build switch, off.
#ifndef ELL_MUL_FAST
#define ELL_MUL_FAST 0
#endif
template <int L>
ELL_HD void fe_mul_wide_plain(u32 (&r)[2 * L], const u32 (&a)[L], const u32 (&b)[L]) {
#if defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL) && ELL_MUL_CHAIN
  if constexpr (L == 6) masm::mulc_wide_6(r, a, b);
  else if constexpr (L == 7) masm::mulc_wide_7(r, a, b);
  else if constexpr (L == 8) masm::mulc_wide_8(r, a, b);
  else if constexpr (L == 12) masm::mulc_wide_12(r, a, b);
  else if constexpr (L == 17) masm::mulc_wide_17(r, a, b);
  else bn_mul_wide<L, L>(r, a, b);
#elif defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL)
  if constexpr (L == 6) masm::mul_wide_6(r, a, b);
  else if constexpr (L == 7) masm::mul_wide_7(r, a, b);
  else if constexpr (L == 8) masm::mul_wide_8(r, a, b);
  else if constexpr (L == 12) masm::mul_wide_12(r, a, b);
  else if constexpr (L == 17) masm::mul_wide_17(r, a, b);
  else bn_mul_wide<L, L>(r, a, b);
#else
  bn_mul_wide<L, L>(r, a, b);
#endif
}
template <int L>
ELL_HD void fe_mul_wide(u32 (&r)[2 * L], const u32 (&a)[L], const u32 (&b)[L]) {
#if defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL) && ELL_MUL_CHAIN && ELL_MUL_FAST
  if constexpr (L == 6 || L == 7 || L == 8 || L == 12 || L == 17) {
    u64 ovf = 0;
    if constexpr (L == 6) masm::mulf_wide_6(r, a, b, ovf);
    else if constexpr (L == 7) masm::mulf_wide_7(r, a, b, ovf);
    else if constexpr (L == 8) masm::mulf_wide_8(r, a, b, ovf);
    else if constexpr (L == 12) masm::mulf_wide_12(r, a, b, ovf);
    else masm::mulf_wide_17(r, a, b, ovf);
    if (ELL_UNLIKELY(ovf != 0)) fe_mul_wide_plain<L>(r, a, b);
  } else {
    bn_mul_wide<L, L>(r, a, b);
  }
#else
  fe_mul_wide_plain<L>(r, a, b);
#endif
}
template <int L>
ELL_HD void fe_sqr_wide(u32 (&r)[2 * L], const u32 (&a)[L]) {
#if defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL)
  u32 off[2 * L];
#if ELL_MUL_CHAIN && ELL_MUL_FAST
  if constexpr (L == 6 || L == 7 || L == 8 || L == 12 || L == 17) {
    u64 ovf = 0;
    if constexpr (L == 6) masm::sqrf_offdiag_6(off, a, ovf);
    else if constexpr (L == 7) masm::sqrf_offdiag_7(off, a, ovf);
    else if constexpr (L == 8) masm::sqrf_offdiag_8(off, a, ovf);
    else if constexpr (L == 12) masm::sqrf_offdiag_12(off, a, ovf);
    else masm::sqrf_offdiag_17(off, a, ovf);
    if (ELL_UNLIKELY(ovf != 0)) {
      if constexpr (L == 6) masm::sqrc_offdiag_6(off, a);
      else if constexpr (L == 7) masm::sqrc_offdiag_7(off, a);
      else if constexpr (L == 8) masm::sqrc_offdiag_8(off, a);
      else if constexpr (L == 12) masm::sqrc_offdiag_12(off, a);
      else masm::sqrc_offdiag_17(off, a);
    }
  } else { bn_sqr_wide<L>(r, a); return; }
#elif ELL_MUL_CHAIN
  if constexpr (L == 6) masm::sqrc_offdiag_6(off, a);
  else if constexpr (L == 7) masm::sqrc_offdiag_7(off, a);
  else if constexpr (L == 8) masm::sqrc_offdiag_8(off, a);
  else if constexpr (L == 12) masm::sqrc_offdiag_12(off, a);
  else if constexpr (L == 17) masm::sqrc_offdiag_17(off, a);
  else { bn_sqr_wide<L>(r, a); return; }
#else
  if constexpr (L == 6) masm::sqr_offdiag_6(off, a);
  else if constexpr (L == 7) masm::sqr_offdiag_7(off, a);
  else if constexpr (L == 8) masm::sqr_offdiag_8(off, a);
  else if constexpr (L == 12) masm::sqr_offdiag_12(off, a);
  else if constexpr (L == 17) masm::sqr_offdiag_17(off, a);
  else { bn_sqr_wide<L>(r, a); return; }
#endif
  // r = 2*off + sum_i a_i^2 2^(64 i): the squares sit in disjoint 64-bit slots
  u32 c = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) {
    u64 d = (u64)a[i] * a[i];
    u32 lo2 = (off[2 * i] << 1) | (i ? (off[2 * i - 1] >> 31) : 0u);
    u32 hi2 = (off[2 * i + 1] << 1) | (off[2 * i] >> 31);
    r[2 * i] = addc32(lo2, (u32)d, c, c);
    r[2 * i + 1] = addc32(hi2, (u32)(d >> 32), c, c);
  }
#else
  bn_sqr_wide<L>(r, a);
#endif
}

// --------------------------------------------------------------------------
// helpers shared by the field families
// --------------------------------------------------------------------------

// r = (a + b) mod p, inputs in [0,p)
template <int L>
ELL_HD void mod_add(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L], const u32 (&p)[L]) {
  u32 t[L], s[L];
  u32 c = bn_add<L>(t, a, b);
  u32 br = bn_sub<L>(s, t, p);
  // result is s when the true sum (c:t) >= p, i.e. when c==1 or no borrow
  bool use_s = (c != 0) || (br == 0);
  bn_select<L>(r, use_s, s, t);
}

template <int L>
ELL_HD void mod_sub(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L], const u32 (&p)[L]) {
  u32 t[L], s[L];
  u32 br = bn_sub<L>(t, a, b);
  bn_add<L>(s, t, p);
  bn_select<L>(r, br != 0, s, t);
}

// Final step of a reduction whose value c*2^(32L) + r is known to be < 2p, for a modulus
// whose top limb is 2^32 - 1 (p within 2^-32 of 2^(32L)): both "c != 0" and "r >= p" are then
// vanishingly rare for every lane of a wave, so the subtraction sits behind a branch that is
// (almost) never taken instead of costing 2L instructions on every multiply.  The result is
// identical to the branch-free form.
template <int L>
ELL_HD void cond_sub_rare(u32 (&r)[L], u32 c, const u32 (&p)[L]) {
  // r >= p needs r's top limb to be all ones (p's is): cheap necessary condition first
  bool maybe = (c != 0) || (r[L - 1] == 0xFFFFFFFFu);
  if (ELL_UNLIKELY(maybe)) {
    u32 s[L];
    u32 br = bn_sub<L>(s, r, p);
    bool take = (c != 0) || (br == 0);
    ELL_UNROLL
    for (int i = 0; i < L; i++) r[i] = take ? s[i] : r[i];
  }
}

// --------------------------------------------------------------------------
// secp256k1 base field
// --------------------------------------------------------------------------
struct FpK256 {
  static constexpr int L = 8;
  typedef Fe<8> El;
  static constexpr u32 C0 = 977u;  // p = 2^256 - 2^32 - 977

  ELL_HD static void get_p(u32 (&p)[8]) {
    p[0] = 0xFFFFFC2Fu; p[1] = 0xFFFFFFFEu;
    ELL_UNROLL
    for (int i = 2; i < 8; i++) p[i] = 0xFFFFFFFFu;
  }
  ELL_HD static El zero() { El r; bn_zero<8>(r.v); return r; }
  ELL_HD static El one() { El r; bn_zero<8>(r.v); r.v[0] = 1; return r; }
  ELL_HD static El from_plain(const u32 (&a)[8]) {        // a < 2^256, reduce once
    u32 p[8]; get_p(p);
    u32 s[8];
    u32 br = bn_sub<8>(s, a, p);
    El r; bn_select<8>(r.v, br == 0, s, a);
    return r;
  }
  ELL_HD static void to_plain(u32 (&r)[8], const El& a) { bn_copy<8>(r, a.v); }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<8>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<8>(a.v, b.v); }
  ELL_HD static bool is_odd(const El& a) { return a.v[0] & 1; }

  // a + b mod p with p = 2^256 - delta, delta = 2^32 + 977: when the 256-bit sum carries out,
  // subtracting p is adding delta to the two low limbs (the result is then < p - 1, no second
  // carry out).  Everything beyond that is rare for every lane of a wave and sits behind one
  // branch: a carry out of limb 1 (needs limb 1 of the sum = 2^32 - 1) that has to ripple
  // through limbs 2..7, and a carry-less sum in [p, 2^256) (needs limb 7 = 2^32 - 1).
  ELL_HD static El add(const El& a, const El& b) {
    u32 t[8];
    u32 c = bn_add<8>(t, a.v, b.v);
    u32 d0 = c ? C0 : 0u, d1 = c;
    u32 cc = 0;
    El r;
    r.v[0] = addc32(t[0], d0, cc, cc);
    r.v[1] = addc32(t[1], d1, cc, cc);
    ELL_UNROLL
    for (int i = 2; i < 8; i++) r.v[i] = t[i];
    if (ELL_UNLIKELY(cc != 0 || t[7] == 0xFFFFFFFFu)) {
      ELL_UNROLL
      for (int i = 2; i < 8; i++) r.v[i] = addc32(t[i], 0, cc, cc);
      if (c == 0) {                       // no fold happened: the sum may lie in [p, 2^256)
        u32 p[8]; get_p(p);
        u32 s[8];
        u32 br = bn_sub<8>(s, r.v, p);
        ELL_UNROLL
        for (int i = 0; i < 8; i++) r.v[i] = br ? r.v[i] : s[i];
      }
    }
    return r;
  }
  // a - b mod p: on borrow, adding p is subtracting delta from the two low limbs; a borrow out
  // of limb 1 (rare) ripples through limbs 2..7.  The result is canonical either way.
  ELL_HD static El sub(const El& a, const El& b) {
    u32 t[8];
    u32 bw = bn_sub<8>(t, a.v, b.v);
    u32 d0 = bw ? C0 : 0u, d1 = bw;
    u32 bb = 0;
    El r;
    r.v[0] = subb32(t[0], d0, bb, bb);
    r.v[1] = subb32(t[1], d1, bb, bb);
    ELL_UNROLL
    for (int i = 2; i < 8; i++) r.v[i] = t[i];
    if (ELL_UNLIKELY(bb != 0)) {
      ELL_UNROLL
      for (int i = 2; i < 8; i++) r.v[i] = subb32(t[i], 0, bb, bb);
    }
    return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  // 2^K * a (K = 1..3) as one shift: the K bits shifted out of limb 7 fold back in as
  // top * delta on the two low limbs.  Eight independent funnel shifts and a two-limb carry
  // chain instead of K eight-limb chains; the ripple and the final subtraction are rare as in add.
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    static_assert(K >= 1 && K <= 3, "small shifts only");
    u32 lo[8];
    u32 top = a.v[7] >> (32 - K);
    lo[0] = a.v[0] << K;
    ELL_UNROLL
    for (int i = 1; i < 8; i++) lo[i] = (a.v[i] << K) | (a.v[i - 1] >> (32 - K));
    El r;
    u32 c = 0;
    r.v[0] = addc32(lo[0], top * C0, c, c);
    r.v[1] = addc32(lo[1], top, c, c);
    ELL_UNROLL
    for (int i = 2; i < 8; i++) r.v[i] = lo[i];
    if (ELL_UNLIKELY(c != 0 || lo[7] == 0xFFFFFFFFu)) {
      ELL_UNROLL
      for (int i = 2; i < 8; i++) r.v[i] = addc32(lo[i], 0, c, c);
      u32 p[8]; get_p(p);
      u32 s[8];
      u32 br = bn_sub<8>(s, r.v, p);
      bool take = (c != 0) || (br == 0);
      ELL_UNROLL
      for (int i = 0; i < 8; i++) r.v[i] = take ? s[i] : r.v[i];
    }
    return r;
  }

  // fold a 512-bit value t[0..16) to [0,p): 2^256 == 2^32 + 977 (mod p).
  // Written with explicit carry chains (addc32) so that gfx950 gets
  // v_add_co/v_addc instead of 64-bit add + move pairs.
#if defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL)
  // Device form of the first fold: x_i = hi_i * 977 + (lo_i + hi_i * 2^32) is ONE multiply-add
  // per limb -- the 64-bit addend pair (lo_i, hi_i) carries both lo and the hi << 32 term --
  // and u = sum x_i 2^(32 i) is one carry chain.  x_i overflows 64 bits only when
  // hi_i >= 2^32 - 977; the multiply-adds' carry-out masks are OR-ed on the scalar unit and a
  // wave that sees one (probability ~1e-4 per multiply) redoes the fold the generic way.
  ELL_HD static El reduce_wide(const u32 (&t)[16]) {
    u64 x[8];
    u64 m0 = masm::fold4(x[0], x[1], x[2], x[3], ((u64)t[8] << 32) | t[0], ((u64)t[9] << 32) | t[1],
                         ((u64)t[10] << 32) | t[2], ((u64)t[11] << 32) | t[3], t[8], t[9], t[10], t[11], C0);
    u64 m1 = masm::fold4(x[4], x[5], x[6], x[7], ((u64)t[12] << 32) | t[4], ((u64)t[13] << 32) | t[5],
                         ((u64)t[14] << 32) | t[6], ((u64)t[15] << 32) | t[7], t[12], t[13], t[14], t[15], C0);
    if (ELL_UNLIKELY((m0 | m1) != 0)) return reduce_wide_generic(t);
    u32 u[10];
    u32 c = 0;
    u[0] = (u32)x[0];
    ELL_UNROLL
    for (int i = 1; i < 8; i++) u[i] = addc32((u32)x[i], (u32)(x[i - 1] >> 32), c, c);
    u[8] = addc32((u32)(x[7] >> 32), 0, c, c);
    u[9] = c;
    return fold_top(u);
  }
#else
  ELL_HD static El reduce_wide(const u32 (&t)[16]) { return reduce_wide_generic(t); }
#endif
  ELL_HD static El reduce_wide_generic(const u32 (&t)[16]) {
    // v = hi * 977: eight independent 32x10-bit products
    u32 pl[8], ph[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) {
      u64 x = (u64)t[8 + i] * C0;
      pl[i] = (u32)x;
      ph[i] = (u32)(x >> 32);
    }
    // w = v + (hi << 32)                      (10 limbs)
    u32 w[10];
    u32 c = 0;
    w[0] = pl[0];
    ELL_UNROLL
    for (int i = 1; i < 8; i++) w[i] = addc32(pl[i], ph[i - 1], c, c);
    w[8] = ph[7] + c;                          // < 2^10 + 1
    c = 0;
    ELL_UNROLL
    for (int i = 1; i < 9; i++) w[i] = addc32(w[i], t[8 + i - 1], c, c);
    w[9] = c;
    // u = lo + w                              (u[8] + u[9]*2^32 = T < 2^34)
    u32 u[10];
    c = 0;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) u[i] = addc32(t[i], w[i], c, c);
    u[8] = addc32(w[8], 0, c, c);
    u[9] = w[9] + c;
    return fold_top(u);
  }
  // second fold: r = u[0..8) + T*977 + (T << 32) for T = u[8] + u[9]*2^32 < 2^34, canonicalised
  ELL_HD static El fold_top(const u32 (&u)[10]) {
    u32 c;
    u64 q = (u64)u[8] * C0;
    u32 t0 = (u32)q;
    u32 t1 = (u32)(q >> 32) + u[9] * C0;       // < 2^10 + 2^12
    u32 cy;
    t1 = addc32(t1, u[8], 0, cy);
    u32 t2 = u[9] + cy;
    El out;
    c = 0;
    out.v[0] = addc32(u[0], t0, c, c);
    out.v[1] = addc32(u[1], t1, c, c);
    out.v[2] = addc32(u[2], t2, c, c);
    ELL_UNROLL
    for (int i = 3; i < 8; i++) out.v[i] = u[i];
    // t2 <= 4, so a carry out of limb 2 is rare, and so is a value in [p, 2^256) (needs limb 7
    // = 2^32 - 1): ripple / subtract p behind one branch.  value = c*2^256 + r < 2^256 + 2^67.
    if (ELL_UNLIKELY(c != 0 || u[7] == 0xFFFFFFFFu)) {
      ELL_UNROLL
      for (int i = 3; i < 8; i++) out.v[i] = addc32(u[i], 0, c, c);
      u32 p[8]; get_p(p);
      u32 s[8];
      u32 br = bn_sub<8>(s, out.v, p);
      bool take = (c != 0) || (br == 0);
      ELL_UNROLL
      for (int i = 0; i < 8; i++) out.v[i] = take ? s[i] : out.v[i];
    }
    return out;
  }
  ELL_HD static El mul(const El& a, const El& b) {
    u32 t[16];
    fe_mul_wide<8>(t, a.v, b.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr(const El& a) {
    u32 t[16];
    fe_sqr_wide<8>(t, a.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  // a^-1 (0 for 0): Bernstein-Yang division steps, safegcd.h
  static ELL_HD_NOINLINE El inv(const El& a) {
    El r;
    SafeGcd<consts::SECP256K1_P>::inv(r.v, a.v);
    return r;
  }
  // a^((p+1)/4): a square root of a when a is a square (p = 3 mod 4).  The exponent is
  // 223 ones, 0, 22 ones, 0000, 11, 00 in binary -- 253 S + 13 M.  (bn.js Red#sqrt,
  // dist/elliptic.js:7180-7190, takes the same power.)
  static constexpr bool HAS_SQRT = true;
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

// --------------------------------------------------------------------------
// GF(2^255 - 19)
// --------------------------------------------------------------------------
struct Fp25519 {
  static constexpr int L = 8;
  typedef Fe<8> El;

  ELL_HD static void get_p(u32 (&p)[8]) {
    p[0] = 0xFFFFFFEDu;
    ELL_UNROLL
    for (int i = 1; i < 7; i++) p[i] = 0xFFFFFFFFu;
    p[7] = 0x7FFFFFFFu;
  }
  ELL_HD static El zero() { El r; bn_zero<8>(r.v); return r; }
  ELL_HD static El one() { El r; bn_zero<8>(r.v); r.v[0] = 1; return r; }
  // canonicalise any 256-bit value
  ELL_HD static El from_plain(const u32 (&a)[8]) {
    // fold bit 255: a = lo255 + 19*b255   (< 2^255 + 19)
    u32 r[8];
    u32 top = a[7] >> 31;
    u32 c = 0;
    r[0] = addc32(a[0], 19u * top, c, c);
    ELL_UNROLL
    for (int i = 1; i < 7; i++) r[i] = addc32(a[i], 0, c, c);
    r[7] = (a[7] & 0x7FFFFFFFu) + c;
    u32 p[8]; get_p(p);
    u32 s[8];
    u32 br = bn_sub<8>(s, r, p);
    El out; bn_select<8>(out.v, br == 0, s, r);
    return out;
  }
  ELL_HD static void to_plain(u32 (&r)[8], const El& a) { bn_copy<8>(r, a.v); }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<8>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<8>(a.v, b.v); }
  ELL_HD static bool is_odd(const El& a) { return a.v[0] & 1; }
  // r = u + add0 for u < 2^255 and a small add0, canonicalised: the carry out of limb 0 and a
  // value in [p, 2^255 + add0) (needs limbs 1..6 all ones) are both rare for every lane of a
  // wave, so the ripple and the final subtraction sit behind one branch.
  ELL_HD static El finish(const u32 (&u)[8], u32 add0) {
    El r;
    u32 c = 0;
    r.v[0] = addc32(u[0], add0, c, c);
    ELL_UNROLL
    for (int i = 1; i < 8; i++) r.v[i] = u[i];
    if (ELL_UNLIKELY(c != 0 || u[7] == 0x7FFFFFFFu)) {
      ELL_UNROLL
      for (int i = 1; i < 8; i++) r.v[i] = addc32(u[i], 0, c, c);
      r = from_plain(r.v);
    }
    return r;
  }
  // a + b < 2^256: fold bit 255 (2^255 == 19); the sum is < 2p, so one fold canonicalises it
  // except for the 19 values in [p, 2^255) (rare, see finish)
  ELL_HD static El add(const El& a, const El& b) {
    u32 t[8];
    bn_add<8>(t, a.v, b.v);
    u32 top = t[7] >> 31;
    t[7] &= 0x7FFFFFFFu;
    return finish(t, top ? 19u : 0u);
  }
  // a - b: on borrow the wrapped difference is v + 2^256 with bit 255 set, and v + p is that
  // minus 2^255 minus 19
  ELL_HD static El sub(const El& a, const El& b) {
    u32 t[8];
    u32 bw = bn_sub<8>(t, a.v, b.v);
    t[7] &= 0x7FFFFFFFu;                         // bit 255 is set exactly when bw is
    El r;
    u32 bb = 0;
    r.v[0] = subb32(t[0], bw ? 19u : 0u, bb, bb);
    ELL_UNROLL
    for (int i = 1; i < 8; i++) r.v[i] = t[i];
    if (ELL_UNLIKELY(bb != 0)) {
      ELL_UNROLL
      for (int i = 1; i < 8; i++) r.v[i] = subb32(t[i], 0, bb, bb);
    }
    return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  // a * k for a one-limb k < 2^24: an 8-step multiply-accumulate chain; the ninth limb and
  // bit 255 fold back in through finish
  ELL_HD static El mul_u32(const El& a, u32 k) {
    u32 lo[8];
    u64 acc = 0;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) {
      acc = (u64)a.v[i] * k + (acc >> 32);
      lo[i] = (u32)acc;
    }
    u32 carry = (u32)(acc >> 32);                 // < 2^24
    u32 top = lo[7] >> 31;
    lo[7] &= 0x7FFFFFFFu;
    return finish(lo, carry * 38u + 19u * top);
  }
  // 2^K * a (K = 1..3) as one shift: bits 255.. of a << K fold back in as 19 * top
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    static_assert(K >= 1 && K <= 3, "small shifts only");
    u32 lo[8];
    u32 top = a.v[7] >> (31 - K);
    lo[0] = a.v[0] << K;
    ELL_UNROLL
    for (int i = 1; i < 8; i++) lo[i] = (a.v[i] << K) | (a.v[i - 1] >> (32 - K));
    lo[7] &= 0x7FFFFFFFu;
    return finish(lo, 19u * top);
  }

#if defined(ELL_HAVE_MUL_ASM) && !defined(ELL_NO_ASM_MUL)
  // Device form: the even high limbs fold as x_j = 38 * hi_2j + (lo_2j, lo_2j+1) -- the four
  // 64-bit addend pairs tile the low half exactly -- and the odd ones as y_j = 38 * hi_2j+1,
  // which land on disjoint limbs (2j+1, 2j+2); u = E + O is then one carry chain.  x_j can
  // overflow 64 bits (probability ~2^-27 per limb): the OR of the carry-out masks sends the
  // wave to the generic fold.
  ELL_HD static El reduce_wide(const u32 (&t)[16]) {
    u64 x[4];
    u64 m = masm::fold4(x[0], x[1], x[2], x[3], ((u64)t[1] << 32) | t[0], ((u64)t[3] << 32) | t[2],
                        ((u64)t[5] << 32) | t[4], ((u64)t[7] << 32) | t[6], t[8], t[10], t[12], t[14], 38u);
    if (ELL_UNLIKELY(m != 0)) return reduce_wide_generic(t);
    u64 y[4];
    ELL_UNROLL
    for (int j = 0; j < 4; j++) y[j] = (u64)t[9 + 2 * j] * 38u;
    u32 u[8];
    u32 c = 0;
    u[0] = (u32)x[0];
    u[1] = addc32((u32)(x[0] >> 32), (u32)y[0], c, c);
    ELL_UNROLL
    for (int j = 1; j < 4; j++) {
      u[2 * j] = addc32((u32)x[j], (u32)(y[j - 1] >> 32), c, c);
      u[2 * j + 1] = addc32((u32)(x[j] >> 32), (u32)y[j], c, c);
    }
    u32 carry = (u32)(y[3] >> 32) + c;          // < 2^7
    u32 top = u[7] >> 31;
    u[7] &= 0x7FFFFFFFu;
    return finish(u, carry * 38u + 19u * top);
  }
#else
  ELL_HD static El reduce_wide(const u32 (&t)[16]) { return reduce_wide_generic(t); }
#endif
  ELL_HD static El reduce_wide_generic(const u32 (&t)[16]) {
    // v = 38*hi: eight independent 32x6-bit products; u = lo + v (9 limbs)
    u32 pl[8], ph[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) {
      u64 x = (u64)t[8 + i] * 38u;
      pl[i] = (u32)x;
      ph[i] = (u32)(x >> 32);
    }
    u32 w[9];
    u32 c = 0;
    w[0] = pl[0];
    ELL_UNROLL
    for (int i = 1; i < 8; i++) w[i] = addc32(pl[i], ph[i - 1], c, c);
    w[8] = ph[7] + c;
    u32 u[8];
    c = 0;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) u[i] = addc32(t[i], w[i], c, c);
    u32 carry = w[8] + c;                       // < 40
    // fold carry*2^256 == carry*38, and bit 255 == 19, in one pass
    u32 top = u[7] >> 31;
    u[7] &= 0x7FFFFFFFu;
    return finish(u, carry * 38u + 19u * top);  // u < 2^255, add0 < 2^11
  }
  ELL_HD static El mul(const El& a, const El& b) {
    u32 t[16];
    fe_mul_wide<8>(t, a.v, b.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr(const El& a) {
    u32 t[16];
    fe_sqr_wide<8>(t, a.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  // z^-1 (0 for 0): Bernstein-Yang division steps, safegcd.h
  static ELL_HD_NOINLINE El inv(const El& z) {
    El r;
    SafeGcd<consts::P25519_P>::inv(r.v, z.v);
    return r;
  }
  // z^((p-5)/8) = z^(2^252 - 3): the building block of the square root of a ratio
  static ELL_HD_NOINLINE El pow22523(const El& z) {
    El z2 = sqr(z);
    El z9 = mul(sqr_n(z2, 2), z);
    El z11 = mul(z9, z2);
    El z2_5_0 = mul(sqr(z11), z9);
    El z2_10_0 = mul(sqr_n(z2_5_0, 5), z2_5_0);
    El z2_20_0 = mul(sqr_n(z2_10_0, 10), z2_10_0);
    El z2_40_0 = mul(sqr_n(z2_20_0, 20), z2_20_0);
    El z2_50_0 = mul(sqr_n(z2_40_0, 10), z2_10_0);
    El z2_100_0 = mul(sqr_n(z2_50_0, 50), z2_50_0);
    El z2_200_0 = mul(sqr_n(z2_100_0, 100), z2_100_0);
    El z2_250_0 = mul(sqr_n(z2_200_0, 50), z2_50_0);
    return mul(sqr_n(z2_250_0, 2), z);
  }
  // sqrt(-1) = 2^((p-1)/4)
  ELL_HD static El sqrt_m1() {
    El r;
    const u32 c[8] = {0x4A0EA0B0u, 0xC4EE1B27u, 0xAD2FE478u, 0x2F431806u,
                      0x3DFBD7A7u, 0x2B4D0099u, 0x4FC1DF0Bu, 0x2B832480u};
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = c[i];
    return r;
  }
  // x with v*x^2 == u, if one exists (p = 5 mod 8): candidate u v^3 (u v^7)^((p-5)/8),
  // multiplied by sqrt(-1) when v x^2 == -u.  Returns false when u/v is not a square.
  ELL_HD static bool sqrt_ratio(El& x, const El& u, const El& v) {
    El v3 = mul(sqr(v), v);
    El v7 = mul(sqr(v3), v);
    El r = mul(mul(u, v3), pow22523(mul(u, v7)));
    El chk = mul(v, sqr(r));
    bool ok1 = eq(chk, u);
    bool ok2 = eq(chk, neg(u));
    El r2 = mul(r, sqrt_m1());
    bn_select<8>(x.v, ok1, r.v, r2.v);
    return ok1 || ok2;
  }
};

// --------------------------------------------------------------------------
// a^e for a compile-time constant exponent: left-to-right sliding window over odd powers
// a, a^3, ..., a^(2^W - 1) (W = 4 up to 8 limbs, 3 above: the table lives in VGPRs).  The
// exponent is the same for every lane, so window scanning is scalar work and the choice of
// table entry is a uniform branch, not divergence.  ~L*32 squarings + L*32/(W+1) multiplies
// against L*32 + popcount(e) for plain square-and-multiply (the NIST primes and group orders
// are mostly one bits).
// --------------------------------------------------------------------------
template <class F, int L>
ELL_HD typename F::El pow_const_window(const typename F::El& a, const u32 (&e)[L]) {
  typedef typename F::El El;
  constexpr int W = L <= 8 ? 4 : 3;
  constexpr int T = 1 << (W - 1);
  El tbl[T];
  tbl[0] = a;
  {
    El a2 = F::sqr(a);
    ELL_UNROLL
    for (int i = 1; i < T; i++) tbl[i] = F::mul(tbl[i - 1], a2);
  }
  El r = F::one();
  bool started = false;
  int i = 32 * L - 1;
  ELL_NOUNROLL
  while (i >= 0) {
    if (!((e[i >> 5] >> (i & 31)) & 1u)) {
      if (started) r = F::sqr(r);
      i--;
      continue;
    }
    int j = i - W + 1 < 0 ? 0 : i - W + 1;
    while (!((e[j >> 5] >> (j & 31)) & 1u)) j++;          // window [i..j] ends in a one bit
    u32 val = 0;
    for (int b = i; b >= j; b--) val = (val << 1) | ((e[b >> 5] >> (b & 31)) & 1u);
    if (started) {
      ELL_NOUNROLL
      for (int b = i; b >= j; b--) r = F::sqr(r);
    }
    El m;
    switch (val >> 1) {                                     // uniform: a scalar branch
      case 0: m = tbl[0]; break;
      case 1: m = tbl[1]; break;
      case 2: m = tbl[2]; break;
      case 3: m = tbl[3]; break;
      case 4: m = tbl[T > 4 ? 4 : 0]; break;
      case 5: m = tbl[T > 4 ? 5 : 0]; break;
      case 6: m = tbl[T > 4 ? 6 : 0]; break;
      default: m = tbl[T > 4 ? 7 : 0]; break;
    }
    r = started ? F::mul(r, m) : m;
    started = true;
    i = j - 1;
  }
  return r;
}

// --------------------------------------------------------------------------
// Generic odd modulus, Montgomery form.  P supplies:
//   static constexpr int L; u32 p[L]; u32 n0 (= -p^-1 mod 2^32);
//   u32 one[L] (= R mod p); u32 r2[L] (= R^2 mod p); u32 pm2[L] (= p - 2)
// --------------------------------------------------------------------------
template <class P>
struct FpMont {
  static constexpr int L = P::L;
  typedef Fe<P::L> El;

  ELL_HD static void get_p(u32 (&p)[L]) {
    ELL_UNROLL
    for (int i = 0; i < L; i++) p[i] = P::p[i];
  }
  ELL_HD static El zero() { El r; bn_zero<L>(r.v); return r; }
  ELL_HD static El one() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < L; i++) r.v[i] = P::one[i];
    return r;
  }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<L>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<L>(a.v, b.v); }
  ELL_HD static El add(const El& a, const El& b) {
    u32 p[L]; get_p(p);
    El r; mod_add<L>(r.v, a.v, b.v, p); return r;
  }
  ELL_HD static El sub(const El& a, const El& b) {
    u32 p[L]; get_p(p);
    El r; mod_sub<L>(r.v, a.v, b.v, p); return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  // 2^K * a (K = 1..3): repeated doubling
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    El r = dbl(a);
    ELL_UNROLL
    for (int i = 1; i < K; i++) r = dbl(r);
    return r;
  }

  // Montgomery reduction of a 2L-limb value: t * R^-1 mod p, row by row with explicit
  // carry chains (L multiplies by the constant limbs of p + 2L add-with-carry per row).
  ELL_HD static El redc(u32 (&t)[2 * L]) {
    u32 top = 0;
    ELL_UNROLL
    for (int i = 0; i < L; i++) {
      u32 m = t[i] * P::n0;
      u32 lo[L], hi[L];
      ELL_UNROLL
      for (int j = 0; j < L; j++) {
        u64 x = (u64)m * P::p[j];
        lo[j] = (u32)x;
        hi[j] = (u32)(x >> 32);
      }
      // u = m*p as L+1 limbs, then t[i .. i+L] += u
      u32 u[L + 1];
      u32 c = 0;
      u[0] = lo[0];
      ELL_UNROLL
      for (int j = 1; j < L; j++) u[j] = addc32(lo[j], hi[j - 1], c, c);
      u[L] = hi[L - 1] + c;
      c = 0;
      ELL_UNROLL
      for (int j = 0; j < L; j++) t[i + j] = addc32(t[i + j], u[j], c, c);
      if (i + L < 2 * L) {
        u32 c1, c2;
        u32 y = addc32(t[i + L], u[L], c, c1);
        t[i + L] = addc32(y, 0, top, c2);
        top = c1 + c2;
      }
    }
    // NB: for i = L-1 the row's top limb is t[2L-1]; `top` is the 2^(64L... ) overflow bit
    u32 p[L]; get_p(p);
    u32 r[L];
    ELL_UNROLL
    for (int i = 0; i < L; i++) r[i] = t[L + i];
    El out;
    if constexpr (P::p[L - 1] == 0xFFFFFFFFu) {
      cond_sub_rare<L>(r, top, p);
      bn_copy<L>(out.v, r);
    } else {
      u32 sres[L];
      u32 br = bn_sub<L>(sres, r, p);
      bn_select<L>(out.v, (top != 0) || (br == 0), sres, r);
    }
    return out;
  }

  // Montgomery product a*b*R^-1 mod p.  Device: wide product from the generated
  // v_mad_u64_u32 blocks, then redc().  Host passes: classic CIOS (2L^2 + L multiplies).
  ELL_HD static El mul_inline(const El& a, const El& b) {
#if defined(ELL_HAVE_MUL_ASM) || defined(ELL_TEST_REDC)
    u32 t[2 * L];
    fe_mul_wide<L>(t, a.v, b.v);
    return redc(t);
#else

    u32 t[L + 2];
    ELL_UNROLL
    for (int i = 0; i < L + 2; i++) t[i] = 0;
    ELL_UNROLL
    for (int i = 0; i < L; i++) {
      u32 carry = 0;
      ELL_UNROLL
      for (int j = 0; j < L; j++) {
        u64 x = (u64)a.v[i] * b.v[j] + t[j] + carry;
        t[j] = (u32)x;
        carry = (u32)(x >> 32);
      }
      u64 y = (u64)t[L] + carry;
      t[L] = (u32)y;
      t[L + 1] = (u32)(y >> 32);
      u32 m = t[0] * P::n0;
      u64 x = (u64)m * P::p[0] + t[0];
      carry = (u32)(x >> 32);
      ELL_UNROLL
      for (int j = 1; j < L; j++) {
        x = (u64)m * P::p[j] + t[j] + carry;
        t[j - 1] = (u32)x;
        carry = (u32)(x >> 32);
      }
      y = (u64)t[L] + carry;
      t[L - 1] = (u32)y;
      t[L] = t[L + 1] + (u32)(y >> 32);
    }
    u32 p[L]; get_p(p);
    u32 r[L], s[L];
    ELL_UNROLL
    for (int i = 0; i < L; i++) r[i] = t[i];
    u32 br = bn_sub<L>(s, r, p);
    El out;
    bn_select<L>(out.v, (t[L] != 0) || (br == 0), s, r);
    return out;
  #endif
  }
  ELL_HD static El sqr_inline(const El& a) {
#if defined(ELL_HAVE_MUL_ASM) || defined(ELL_TEST_REDC)
    u32 t[2 * L];
    fe_sqr_wide<L>(t, a.v);
    return redc(t);
#else
    return mul_inline(a, a);
#endif
  }
  // Wide moduli (>= 12 limbs): the multiply is a real function call.  Fully inlined, a
  // p384/p521 point addition keeps ~10 field elements plus the product rows live and the
  // allocator ends at 256 VGPRs + AGPR spills (1 wave/SIMD); as a call the multiply's
  // temporaries die at its return.
  static ELL_HD_NOINLINE El mul_call(El a, El b) { return mul_inline(a, b); }
  static ELL_HD_NOINLINE El sqr_call(El a) { return sqr_inline(a); }
  ELL_HD static El mul(const El& a, const El& b) {
    if constexpr (L >= ELL_MONT_CALL_MINL) return mul_call(a, b);
    else return mul_inline(a, b);
  }
  ELL_HD static El sqr(const El& a) {
    if constexpr (L >= ELL_MONT_CALL_MINL) return sqr_call(a);
    else return sqr_inline(a);
  }

  ELL_HD static El from_plain(const u32 (&a)[L]) {      // a < 2^(32L): a*R mod p
    El x, r2;
    // a may be >= p (scalars are not reduced by the caller): the Montgomery
    // product only needs a < R and b < p for a fully reduced result
    bn_copy<L>(x.v, a);
    ELL_UNROLL
    for (int i = 0; i < L; i++) r2.v[i] = P::r2[i];
    return mul(x, r2);
  }
  ELL_HD static void to_plain(u32 (&r)[L], const El& a) {
    El o; bn_zero<L>(o.v); o.v[0] = 1;
    El x = mul(a, o);
    bn_copy<L>(r, x.v);
  }
  ELL_HD static bool is_odd(const El& a) {
    u32 r[L]; to_plain(r, a); return r[0] & 1;
  }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  // a^((p+1)/4): square root for p = 3 (mod 4) (bn.js Red#sqrt takes the same power)
  static constexpr bool HAS_SQRT = P::P3MOD4;
  static ELL_HD_NOINLINE El sqrt(const El& a) { return pow_const_window<FpMont<P>, L>(a, P::pp1d4); }
  // (aR)^-1 by division steps (safegcd.h) is a^-1 R^-1; one Montgomery multiply by R^3 gives
  // a^-1 R.  (Amortised over a whole batch by Montgomery's trick, see normalize.)
  static ELL_HD_NOINLINE El inv(const El& a) {
    El y, r3;
    SafeGcd<P>::inv(y.v, a.v);
    ELL_UNROLL
    for (int i = 0; i < L; i++) r3.v[i] = P::r3[i];
    return mul(y, r3);
  }
};

// --------------------------------------------------------------------------
// NIST primes of generalised-Mersenne (Solinas) form: plain residues, the 2L-word product is
// folded with the rule B^L == +-B^j ... that the prime's shape gives (additions only, no
// multiplies; a Montgomery reduction spends another L^2 multiplies here).
// R supplies: MP (constant struct with p, pm2, pp1d4, the inversion constants) and the rule
//   NFOLD, fold_pos[NFOLD] (highest first), fold_sign[NFOLD].
// --------------------------------------------------------------------------
// ELL_CHAIN_SCHED = 1: the chains of reduce_wide_chain are kept apart from each other and from
// the product by scheduling barriers (the scheduler otherwise runs three chains side by side and
// pulls them up into the product: more live words)
#ifndef ELL_CHAIN_SCHED
#define ELL_CHAIN_SCHED 0
#endif
#ifndef ELL_P384_CHAIN
#define ELL_P384_CHAIN 0
#endif
#ifndef ELL_SOLINAS_PAIR
#define ELL_SOLINAS_PAIR 1
#endif
#ifndef ELL_P384_PAIR
#define ELL_P384_PAIR 0
#endif
#ifndef ELL_SOLINAS_DIRECT_SUB
#define ELL_SOLINAS_DIRECT_SUB 1
#endif
#if ELL_CHAIN_SCHED && defined(__HIP_DEVICE_COMPILE__)
#define ELL_CHAIN_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define ELL_CHAIN_FENCE() ((void)0)
#endif
template <class R>
struct FpSolinas {
  typedef typename R::MP MP;
  static constexpr int L = MP::L;
  typedef Fe<MP::L> El;
  // p = 3 (mod 4): one exponentiation; p224 (the only other prime here, L == 7): Tonelli-Shanks
  static constexpr bool HAS_SQRT = MP::P3MOD4 || L == 7;

  ELL_HD static void get_p(u32 (&p)[L]) {
    ELL_UNROLL
    for (int i = 0; i < L; i++) p[i] = MP::p[i];
  }
  ELL_HD static El zero() { El r; bn_zero<L>(r.v); return r; }
  ELL_HD static El one() { El r; bn_zero<L>(r.v); r.v[0] = 1; return r; }
  ELL_HD static El from_plain(const u32 (&a)[L]) {        // a < 2^(32L) < 2p
    u32 p[L]; get_p(p);
    u32 s[L];
    u32 br = bn_sub<L>(s, a, p);
    El r; bn_select<L>(r.v, br == 0, s, a);
    return r;
  }
  ELL_HD static void to_plain(u32 (&r)[L], const El& a) { bn_copy<L>(r, a.v); }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<L>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<L>(a.v, b.v); }
  ELL_HD static bool is_odd(const El& a) { return a.v[0] & 1; }
  // 2^(32L) mod p as L little-endian words: the fold rule B^L == sum_f sign_f B^pos_f read as a
  // number (p384: B^4 + B^3 - B + 1 = {1, 2^32-1, 2^32-1, 0, 1, 0...}; p256: {1, 0, 0, 2^32-1,
  // 2^32-1, 2^32-1, 2^32-2, 0}).  a + b that carries out of 32L bits is reduced by adding these
  // words (a - b that borrows by subtracting them), as FpK256::add / sub do with 2^32 + 977 --
  // and only where they are not zero: a carry that would have to travel through a limb whose
  // word is zero after a SMALL word was added (or through a second zero limb in a row) needs
  // that limb at 2^32 - 1 (0 for a borrow), and a carry-less sum in [p, 2^(32L)) needs a top
  // limb of 2^32 - 1; those lanes take a rarely entered branch that redoes the operation the
  // generic way.  p192 10 instead of 18 instructions, p224 14 / 21, p256 16 / 24, p384 20 / 36.
  struct CWords { u32 w[L]; };
  static constexpr CWords cwords() {
    long long acc[L] = {};
    for (int f = 0; f < R::NFOLD; f++) acc[R::fold_pos[f]] += R::fold_sign[f];
    CWords c{};
    long long carry = 0;
    for (int k = 0; k < L; k++) {
      long long v = acc[k] + carry;
      long long q = v >= 0 ? v / 4294967296LL : -((-v + 4294967295LL) / 4294967296LL);
      c.w[k] = (u32)(v - q * 4294967296LL);
      carry = q;
    }
    return c;
  }
  // what limb k does in the masked add / sub of the words: 2 = the word itself joins the chain,
  // 1 = zero word, but the limb below added a large word (its carry is common): chain continues,
  // 0 = zero word and a carry into it is rare: the limb is copied, an arriving carry is flagged
  static constexpr int limb_mode(int k) {
    constexpr CWords C = cwords();
    if (C.w[k] != 0) return 2;
    if (k > 0 && C.w[k - 1] >= 0x10000u) return 1;
    return 0;
  }
  ELL_HD static El add(const El& a, const El& b) {
    constexpr CWords C = cwords();
    u32 t[L];
    u32 c = bn_add<L>(t, a.v, b.v);
    El r;
    u32 cc = 0, rare = (t[L - 1] == 0xFFFFFFFFu) ? 1u : 0u;
    ELL_UNROLL
    for (int k = 0; k < L; k++) {
      const int m = limb_mode(k);
      if (m == 2) r.v[k] = addc32(t[k], c ? C.w[k] : 0u, cc, cc);
      else if (m == 1) r.v[k] = addc32(t[k], 0u, cc, cc);
      else { rare |= cc; cc = 0; r.v[k] = t[k]; }
    }
    rare |= cc;                                   // (cannot happen for canonical inputs; harmless)
    if (ELL_UNLIKELY(rare != 0)) {
      u32 p[L]; get_p(p);
      mod_add<L>(r.v, a.v, b.v, p);
    }
    return r;
  }
  ELL_HD static El sub(const El& a, const El& b) {
    constexpr CWords C = cwords();
    u32 t[L];
    u32 bw = bn_sub<L>(t, a.v, b.v);
    El r;
    u32 bb = 0, rare = 0;
    ELL_UNROLL
    for (int k = 0; k < L; k++) {
      const int m = limb_mode(k);
      if (m == 2) r.v[k] = subb32(t[k], bw ? C.w[k] : 0u, bb, bb);
      else if (m == 1) r.v[k] = subb32(t[k], 0u, bb, bb);
      else { rare |= bb; bb = 0; r.v[k] = t[k]; }
    }
    rare |= bb;
    if (ELL_UNLIKELY(rare != 0)) {
      u32 p[L]; get_p(p);
      mod_sub<L>(r.v, a.v, b.v, p);
    }
    return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  // 2^K * a (K = 1..3): repeated doubling
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    El r = dbl(a);
    ELL_UNROLL
    for (int i = 1; i < K; i++) r = dbl(r);
    return r;
  }

  // Fold of the 2L-word product with 64-bit lazy accumulators: B^L == sum_f sign_f B^pos_f
  // (B = 2^32; R::fold_pos / fold_sign), so the word at position k >= L is added to (or
  // subtracted from) the accumulators at k - L + pos_f, top word first; accumulators are signed
  // 64-bit (a 64-bit add has no carry flag and no carry hazard on gfx950) and stay below 2^44.
  // One carry propagation brings them back to L words plus a small signed carry, which is
  // folded the same way into the lowest max(pos)+1 words; a ripple beyond those and the final
  // range correction are rare (they need a word equal to 0 / 2^32-1) and share one branch.
  // Same value as the FIPS 186-4 D.2 word sums, ~45 % fewer carry-class instructions.
  // acc + w for a 32-bit word w: ONE v_mad_u64_u32 (w * 1 + acc) instead of building the 64-bit
  // pair (w, 0) with two moves and adding it -- the product words never become 64-bit values
  ELL_HD static i64 add_word(i64 acc, u32 w) {
#if defined(__HIP_DEVICE_COMPILE__) && ELL_SOLINAS_MAD_FOLD
    u64 out, sd;
    asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(out), "=s"(sd) : "v"(w), "v"((u64)acc));
    (void)sd;
    return (i64)out;
#else
    return acc + (i64)(u64)w;
#endif
  }
  // x >> 32 (arithmetic) as one v_ashrrev_i64 instead of a 32-bit shift plus a move into a pair
  ELL_HD static i64 sar32(i64 x) {
#if defined(__HIP_DEVICE_COMPILE__) && ELL_SOLINAS_MAD_FOLD
    i64 out;
    asm("v_ashrrev_i64 %0, 32, %1" : "=v"(out) : "v"(x));
    return out;
#else
    return x >> 32;
#endif
  }
  // ---- chain fold (R::CHAIN: p192, p224, p384 -- the primes whose fold rule reaches at most
  // half way up, B^L == sum_f sign_f B^pos_f with pos_0 <= L/2) --------------------------------
  // With T = Lo + H B^L:   V = Lo + sum_f sign_f H B^pos_f   is NFOLD multi-word add / subtract
  // chains of L words each (carry flags, no 64-bit widening, no negation), V < 2 B^(L + pos_0).
  // Its excess G = V >> 32L has only pos_0 + 1 words, so the same step once more,
  // V' = V_lo + sum_f sign_f G B^pos_f, is NFOLD chains of pos_0 + 1 words -- and V' < B^L.
  // Every chain carries one word past its operand; a carry / borrow that wants to travel FURTHER
  // needs that word to be 2^32 - 1 (0): those lanes (~2^-29 per product) finish the ripple in a
  // rarely entered branch right behind the chain (nothing of the product has to stay live for a
  // fallback), and a V' outside [0, p) -- a carry out of the top, or a top word of 2^32 - 1 --
  // takes the final correction branch.  p384: ~90 instructions instead
  // of ~150 (55 + 24 chain words instead of 48 64-bit accumulations + 48 widening moves + 24
  // negation words + the carry pass).  Additions first, subtractions last: V and V' never dip
  // below zero on the way (H B^4 >= H B, H B^3 >= H).
  // One chain.  Its still-pending carry (0 / 1) is shifted into `pend` (pend = 2 pend + c: one
  // v_addc), so that after all 2 NFOLD chains bit (2 NFOLD - 1 - i) of pend belongs to chain i.
  template <bool SUB, int N, int OFF, int NV>
  ELL_HD static void chain(u32 (&v)[NV], const u32* src, u32& pend) {
    ELL_CHAIN_FENCE();
    u32 c = 0;
    ELL_UNROLL
    for (int k = 0; k < N; k++) {
      if (SUB) v[OFF + k] = subb32(v[OFF + k], src[k], c, c);
      else v[OFF + k] = addc32(v[OFF + k], src[k], c, c);
    }
    if constexpr (OFF + N < NV) {                         // one word past the operand
      if (SUB) v[OFF + N] = subb32(v[OFF + N], 0u, c, c);
      else v[OFF + N] = addc32(v[OFF + N], 0u, c, c);
    }
    u32 dummy;
    pend = addc32(pend, pend, c, dummy);
  }
  template <int STAGE, bool SUB, int F, int NV>
  ELL_HD static void chain_folds(u32 (&v)[NV], const u32* src, u32& pend) {
    if constexpr (F < R::NFOLD) {
      constexpr int N = STAGE == 1 ? L : R::fold_pos[0] + 1;
      if constexpr ((R::fold_sign[F] < 0) == SUB) chain<SUB, N, R::fold_pos[F], NV>(v, src, pend);
      chain_folds<STAGE, SUB, F + 1, NV>(v, src, pend);
    }
  }
  // the pending carries as signed corrections: chain i's carry has the weight
  // B^(pos + N + 1) -- B^(pos + N) where the chain had no extra word -- in its stage's frame
  template <int STAGE, bool SUB, int F, int IDX>
  ELL_HD static void chain_pending(i64* A, u32 pend) {
    if constexpr (F < R::NFOLD) {
      if constexpr ((R::fold_sign[F] < 0) == SUB) {
        constexpr int N = STAGE == 1 ? L : R::fold_pos[0] + 1;
        constexpr int NV = STAGE == 1 ? L + R::fold_pos[0] + 1 : L;
        constexpr int E = R::fold_pos[F] + N + (R::fold_pos[F] + N < NV ? 1 : 0);
        const i64 c = (pend >> (2 * R::NFOLD - 1 - IDX)) & 1u;
        A[E] += SUB ? -c : c;
        chain_pending<STAGE, SUB, F + 1, IDX + 1>(A, pend);
      } else {
        chain_pending<STAGE, SUB, F + 1, IDX>(A, pend);
      }
    }
  }
  static constexpr int n_pos() {
    int n = 0;
    for (int f = 0; f < R::NFOLD; f++) n += R::fold_sign[f] > 0 ? 1 : 0;
    return n;
  }
  ELL_HD static El reduce_wide_chain(const u32 (&t)[2 * L]) {
    constexpr int P = R::fold_pos[0];                     // positions are listed highest first
    static_assert(2 * P <= L, "chain fold needs the fold rule to reach at most half way up");
    constexpr int NV = L + P + 1;
    u32 v[NV], h[L];
    ELL_UNROLL
    for (int k = 0; k < NV; k++) v[k] = k < L ? t[k] : 0u;
    ELL_UNROLL
    for (int k = 0; k < L; k++) h[k] = t[L + k];
    u32 pend = 0;
    chain_folds<1, false, 0, NV>(v, h, pend);
    chain_folds<1, true, 0, NV>(v, h, pend);
    ELL_CHAIN_FENCE();
    u32 g[P + 1], r[L];
    ELL_UNROLL
    for (int k = 0; k <= P; k++) g[k] = v[L + k];
    ELL_UNROLL
    for (int k = 0; k < L; k++) r[k] = v[k];
    chain_folds<2, false, 0, L>(r, g, pend);
    chain_folds<2, true, 0, L>(r, g, pend);
    if (ELL_UNLIKELY(pend != 0 || r[L - 1] == 0xFFFFFFFFu)) {
      // V' + the carries that were left pending, as a word vector for the lazy fold: stage-2
      // carries sit at their own position, a stage-1 carry at B^e with e >= L is a word of the
      // excess and folds like one.  (Stage 1's words above L were consumed as G already: only the
      // corrections remain there.)
      constexpr int NP = n_pos();
      i64 A[2 * L];
      ELL_UNROLL
      for (int k = 0; k < 2 * L; k++) A[k] = k < L ? (i64)(u64)r[k] : 0;
      chain_pending<1, false, 0, 0>(A, pend);
      chain_pending<1, true, 0, NP>(A, pend);
      chain_pending<2, false, 0, R::NFOLD>(A, pend);
      chain_pending<2, true, 0, R::NFOLD + NP>(A, pend);
      return finish_lazy(A);
    }
    El out;
    bn_copy<L>(out.v, r);
    return out;
  }
  template <class RR, class = void>
  struct has_chain { static constexpr bool value = false; };
  template <class RR>
  struct has_chain<RR, decltype((void)RR::CHAIN)> { static constexpr bool value = RR::CHAIN; };
  ELL_HD static El reduce_wide(const u32 (&t)[2 * L]) {
    if constexpr (has_chain<R>::value && ELL_SOLINAS_CHAIN) return reduce_wide_chain(t);
    else return reduce_wide_lazy(t);
  }
  ELL_HD static El reduce_wide_lazy(const u32 (&t)[2 * L]) {
#if ELL_SOLINAS_MAD_FOLD
    // A[k] holds only what has been folded INTO position k; the product word t[k] itself joins
    // through add_word where the position is consumed
    i64 A[2 * L];
    ELL_UNROLL
    for (int k = 0; k < 2 * L; k++) A[k] = 0;
    ELL_UNROLL
    for (int k = 2 * L - 1; k >= L; k--) {
      const i64 v = add_word(A[k], t[k]);
      ELL_UNROLL
      for (int f = 0; f < R::NFOLD; f++) {
        if (R::fold_sign[f] > 0) A[k - L + R::fold_pos[f]] += v;
        else A[k - L + R::fold_pos[f]] -= v;
      }
    }
    u32 r[L];
    i64 c = 0;
    ELL_UNROLL
    for (int k = 0; k < L; k++) {
      i64 sum = add_word(A[k] + c, t[k]);
      r[k] = (u32)sum;
      c = sar32(sum);
    }
    return finish_top(r, c);
#else
    i64 A[2 * L];
    ELL_UNROLL
    for (int k = 0; k < 2 * L; k++) A[k] = (i64)(u64)t[k];
    return finish_lazy(A);
#endif
  }
  // the lazy-accumulator fold of a signed word vector A[0 .. 2L)
  ELL_HD static El finish_lazy(i64 (&A)[2 * L]) {
    ELL_UNROLL
    for (int k = 2 * L - 1; k >= L; k--) {
      const i64 v = A[k];
      if constexpr (R::NFOLD - n_pos() == 1 && ELL_SOLINAS_DIRECT_SUB) {
        // a single subtracted position (p384, p224): subtract v there (two instructions) instead
        // of negating it first (two) and adding (one)
        ELL_UNROLL
        for (int f = 0; f < R::NFOLD; f++) {
          if (R::fold_sign[f] > 0) A[k - L + R::fold_pos[f]] += v;
          else A[k - L + R::fold_pos[f]] -= v;
        }
      } else {
        const i64 nv = -v;
        ELL_UNROLL
        for (int f = 0; f < R::NFOLD; f++) A[k - L + R::fold_pos[f]] += R::fold_sign[f] > 0 ? v : nv;
      }
    }
    u32 r[L];
    i64 c = 0;
    ELL_UNROLL
    for (int k = 0; k < L; k++) {
      i64 sum = A[k] + c;
      r[k] = (u32)sum;
      c = sum >> 32;
    }
    return finish_top(r, c);
  }
  ELL_HD static El finish_top(u32 (&r)[L], i64 c) {
    // c * B^L: the same fold on the low words
    constexpr int TOP = R::fold_pos[0] + 1;            // fold_pos is listed highest first
    i64 c2 = 0;
    ELL_UNROLL
    for (int k = 0; k < TOP; k++) {
      i64 e = 0;
      ELL_UNROLL
      for (int f = 0; f < R::NFOLD; f++)
        if (R::fold_pos[f] == k) e += R::fold_sign[f] > 0 ? c : -c;
      i64 sum = (i64)(u64)r[k] + e + c2;
      r[k] = (u32)sum;
      c2 = sar32(sum);
    }
    if (ELL_UNLIKELY(c2 != 0 || r[L - 1] == 0xFFFFFFFFu)) {
      ELL_UNROLL
      for (int k = TOP; k < L; k++) {
        i64 sum = (i64)(u64)r[k] + c2;
        r[k] = (u32)sum;
        c2 = sar32(sum);
      }
      // value = c2 * B^L + r with |c2| <= 1: bring it into [0, p)
      u32 p[L]; get_p(p);
      ELL_NOUNROLL
      for (int it = 0; it < 3; it++) {
        u32 sm[L], sp[L];
        u32 bs = bn_sub<L>(sm, r, p);
        u32 ca = bn_add<L>(sp, r, p);
        bool neg = c2 < 0;
        bool big = c2 > 0 || (c2 == 0 && bs == 0);
        ELL_UNROLL
        for (int i = 0; i < L; i++) r[i] = neg ? sp[i] : (big ? sm[i] : r[i]);
        c2 = neg ? c2 + (i64)ca : (big ? c2 - (i64)bs : c2);
      }
    }
    El out;
    bn_copy<L>(out.v, r);
    return out;
  }
  ELL_HD static El mul_inline(const El& a, const El& b) {
    u32 t[2 * L];
    fe_mul_wide<L>(t, a.v, b.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr_inline(const El& a) {
    u32 t[2 * L];
    fe_sqr_wide<L>(t, a.v);
    return reduce_wide(t);
  }
  static ELL_HD_NOINLINE El mul_call(El a, El b) { return mul_inline(a, b); }
  static ELL_HD_NOINLINE El sqr_call(El a) { return sqr_inline(a); }
  ELL_HD static El mul(const El& a, const El& b) {
    if constexpr (L >= ELL_MONT_CALL_MINL) return mul_call(a, b);
    else return mul_inline(a, b);
  }
  ELL_HD static El sqr(const El& a) {
    if constexpr (L >= ELL_MONT_CALL_MINL) return sqr_call(a);
    else return sqr_inline(a);
  }
  // PAIR PRODUCTS with one fold (round 5, ELL_SOLINAS_PAIR): the lazy fold takes a SIGNED word
  // vector, so a difference of two wide products -- Y3 = rr (v - X3) - Y1 hhh of the mixed
  // addition, Y3 = alpha (4 beta - X3) - 8 gamma^2 of the a = -3 doubling (short.h) -- is folded
  // once instead of twice.  Only where the lazy fold is the field's fold (p256, p384).
  // MEASURED (MI355X, same box, profiles/r05_pair_products_ab.txt): p256 P*k 61.7 -> 64.6 M/s
  // (+4.7 %, kernel 16.75 -> 15.98 ms per 2^20); p384 21.65 -> 21.3-21.5 M/s with one pass in
  // flight (kernel 11.92 -> 12.0-12.1 ms per 2^18: the second wide product's 24 words live across
  // the fold cost the 168-register kernel more than the fold saves) and +0.7 % with two -- so the
  // prime's descriptor decides (R::PAIR: p256 yes, p384 no).
  template <class RR, class = void>
  struct wants_pair { static constexpr bool value = false; };
  template <class RR>
  struct wants_pair<RR, decltype((void)RR::PAIR)> { static constexpr bool value = RR::PAIR; };
  static constexpr bool PAIR = ELL_SOLINAS_PAIR && wants_pair<R>::value && !(has_chain<R>::value && ELL_SOLINAS_CHAIN);
  // a * b - c * d
  ELL_HD static El mul_sub_mul(const El& a, const El& b, const El& c, const El& d) {
    u32 t1[2 * L], t2[2 * L];
    fe_mul_wide<L>(t1, a.v, b.v);
    fe_mul_wide<L>(t2, c.v, d.v);
    i64 A[2 * L];
    ELL_UNROLL
    for (int k = 0; k < 2 * L; k++) A[k] = (i64)(u64)t1[k] - (i64)(u64)t2[k];
    return finish_lazy(A);
  }
  // a * b - 8 c^2
  ELL_HD static El mul_sub_sqr8(const El& a, const El& b, const El& c) {
    u32 t1[2 * L], t2[2 * L];
    fe_mul_wide<L>(t1, a.v, b.v);
    fe_sqr_wide<L>(t2, c.v);
    i64 A[2 * L];
    ELL_UNROLL
    for (int k = 0; k < 2 * L; k++) A[k] = (i64)(u64)t1[k] - ((i64)(u64)t2[k] << 3);
    return finish_lazy(A);
  }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  static ELL_HD_NOINLINE El inv(const El& a) {
    El r;
    SafeGcd<MP>::inv(r.v, a.v);
    return r;
  }
  // Square root modulo p224 = 2^96 (2^128 - 1) + 1, where bn.js takes its generic Tonelli-Shanks
  // loop (Red#sqrt, dist/elliptic.js:7259-7311).  Same algorithm with the data-dependent inner
  // search replaced by a fixed schedule, so that all lanes run the same instructions: with
  // q = 2^128 - 1, x = a^((q+1)/2), b = a^q and c = 11^q (11 = the least non-residue; c generates the
  // 2^96 roots of unity), the discrete logarithm of b is cleared from the bottom while x^2 == a b
  // is kept: four bits per step from tables (default, below), or -- ELL_P224_TS_WINDOW = 0, the
  // first version -- one bit per step with c squared along (4.5 k squarings).  For a non-residue
  // the result is garbage and the caller's x^2 == a test fails (the reference: 'Assertion failed'
  // out of the loop's assert(i < m)).  Any root will do: every caller fixes the parity afterwards.
  ELL_HD static El sqrt_ts224(const El& a) {
    static_assert(L == 7 || MP::P3MOD4, "Tonelli-Shanks constants are p224's");
    // a^(2^127 - 1): 126 S + 12 M over exponents 2^n - 1
    El e1 = a;
    El e2 = mul(sqr(e1), e1);
    El e3 = mul(sqr(e2), e1);
    El e6 = mul(sqr_n(e3, 3), e3);
    El e7 = mul(sqr(e6), e1);
    El e14 = mul(sqr_n(e7, 7), e7);
    El e15 = mul(sqr(e14), e1);
    El e30 = mul(sqr_n(e15, 15), e15);
    El e31 = mul(sqr(e30), e1);
    El e62 = mul(sqr_n(e31, 31), e31);
    El e63 = mul(sqr(e62), e1);
    El e126 = mul(sqr_n(e63, 63), e63);
    El t = mul(sqr(e126), e1);
    El x = mul(a, t);
    El b = mul(x, t);
#if ELL_P224_TS_WINDOW
    // Windowed form (round 2): with g = c^2 (order 2^95) the logarithm e of b to the base g is
    // found four bits at a time from the bottom -- b_j^(2^(91-4j)) = H[e_j] for the sixteen
    // powers H of g^(2^91) -- and digit j is removed by x *= CN[j][e_j] = c^(-16^j e_j),
    // b *= CN[j][e_j]^2 (consts::P224_TS, tools/gen_consts.py).  1 081 squarings instead of 4 465;
    // the table entry is picked per lane by compare-and-select, every lane runs the same code.
    typedef consts::P224_TS TS;
    ELL_NOUNROLL
    for (int j = 0; j < TS::WINDOWS; j++) {
      El w = b;
      if (j < TS::WINDOWS - 1) w = sqr_n(b, 91 - 4 * j);
      // last window: three bits left, b itself = H[2 e_j]
      const int step = (j < TS::WINDOWS - 1) ? 1 : 2;
      u32 idx = 0;
      ELL_NOUNROLL
      for (int i = 1; i * step < 16; i++) {
        u32 d = 0;
        ELL_UNROLL
        for (int l = 0; l < 7 && l < L; l++) d |= w.v[l] ^ TS::H[(i * step) * 7 + l];
        idx = d == 0 ? (u32)i : idx;
      }
      El f = zero();
      ELL_NOUNROLL
      for (int i = 0; i < 16; i++) {
        ELL_UNROLL
        for (int l = 0; l < 7 && l < L; l++) f.v[l] = idx == (u32)i ? TS::CN[(j * 16 + i) * 7 + l] : f.v[l];
      }
      x = mul(x, f);
      b = mul(b, sqr(f));
    }
#else
    El c = zero();
    constexpr u32 C0[7] = {0xDC691B74u, 0xF3FB3632u, 0xBEA3D8CEu, 0x0B2D6FFBu, 0x0C55B2D4u, 0x8598A792u, 0x6A0FEC67u};
    ELL_UNROLL
    for (int i = 0; i < 7 && i < L; i++) c.v[i] = C0[i];
    const El o = one();
    ELL_NOUNROLL
    for (int k = 0; k <= 94; k++) {
      El w = sqr_n(b, 94 - k);
      bool hit = !eq(w, o);
      El c2 = sqr(c);
      El xc = mul(x, c);
      El bc = mul(b, c2);
      ELL_UNROLL
      for (int i = 0; i < L; i++) {
        x.v[i] = hit ? xc.v[i] : x.v[i];
        b.v[i] = hit ? bc.v[i] : b.v[i];
      }
      c = c2;
    }
#endif
    return x;
  }
  static ELL_HD_NOINLINE El sqrt(const El& a) {
    if constexpr (MP::P3MOD4) return pow_const_window<FpSolinas<R>, L>(a, MP::pp1d4);
    else return sqrt_ts224(a);
  }
};

// fold rules B^L == sum_f fold_sign[f] * B^fold_pos[f]  (B = 2^32), positions listed highest first;
// every one of these primes has 2^32 - 1 as its top word (FpSolinas's rare-branch test)
struct SolP192 {                       // p192 = 2^192 - 2^64 - 1:            B^6 == B^2 + 1
  typedef consts::P192_P MP;
  static constexpr bool CHAIN = true;
  static constexpr int NFOLD = 2;
  static constexpr int fold_pos[2] = {2, 0};
  static constexpr int fold_sign[2] = {1, 1};
};
struct SolP224 {                       // p224 = 2^224 - 2^96 + 1:            B^7 == B^3 - 1
  typedef consts::P224_P MP;
  static constexpr bool CHAIN = true;
  static constexpr int NFOLD = 2;
  static constexpr int fold_pos[2] = {3, 0};
  static constexpr int fold_sign[2] = {1, -1};
};
struct SolP256 {                       // p256 = 2^256 - 2^224 + 2^192 + 2^96 - 1:   B^8 == B^7 - B^6 - B^3 + 1
  typedef consts::P256_P MP;
  static constexpr bool PAIR = true;     // Y3 as ONE folded difference of two wide products (FpSolinas::mul_sub_mul)
  static constexpr int NFOLD = 4;
  static constexpr int fold_pos[4] = {7, 6, 3, 0};
  static constexpr int fold_sign[4] = {1, -1, -1, 1};
};
struct SolP384 {                       // p384 = 2^384 - 2^128 - 2^96 + 2^32 - 1:    B^12 == B^4 + B^3 - B + 1
  typedef consts::P384_P MP;
  // the chain fold is 15 % fewer instructions here too (388 instead of 455 per multiplication) and
  // measures 2 % SLOWER (P*k 21.0 against 21.4 M/s, profiles/r03_solinas_chain_ab.txt): four
  // 13-word carry chains are latency, where the lazy fold's 64-bit adds have no flag to wait for
  static constexpr bool CHAIN = ELL_P384_CHAIN;
  // pair products (one fold for Y3): measured slower here, see FpSolinas::PAIR
  static constexpr bool PAIR = ELL_P384_PAIR;
  static constexpr int NFOLD = 4;
  static constexpr int fold_pos[4] = {4, 3, 1, 0};
  static constexpr int fold_sign[4] = {1, 1, -1, 1};
};

// --------------------------------------------------------------------------
// GF(2^521 - 1) (NIST P-521): Mersenne prime, plain residues in 17 limbs.  The 1042-bit
// product folds as lo521 + (N >> 521): two shifted add chains, no multiplies.
// --------------------------------------------------------------------------
struct FpP521 {
  typedef consts::P521_P MP;
  static constexpr int L = 17;
  typedef Fe<17> El;
  static constexpr bool HAS_SQRT = true;

  ELL_HD static void get_p(u32 (&p)[17]) {
    ELL_UNROLL
    for (int i = 0; i < 16; i++) p[i] = 0xFFFFFFFFu;
    p[16] = 0x1FFu;
  }
  ELL_HD static El zero() { El r; bn_zero<17>(r.v); return r; }
  ELL_HD static El one() { El r; bn_zero<17>(r.v); r.v[0] = 1; return r; }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<17>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<17>(a.v, b.v); }
  ELL_HD static bool is_odd(const El& a) { return a.v[0] & 1; }
  ELL_HD static void to_plain(u32 (&r)[17], const El& a) { bn_copy<17>(r, a.v); }
  // Mersenne add / sub: 2^521 == 1, so bit 521 of a + b is added back at limb 0 (a + b < 2^522
  // fits the 17 limbs), and a borrow of a - b is repaid by subtracting 1 at limb 0 and adding
  // 2^521 at limb 16 (the 2^544 of the wrapped difference leaves through the top).  The carry
  // out of limb 0 (needs limb 0 = 2^32 - 1, or 0 for the borrow) and the sum equal to p itself
  // (all ones) are rare and share one branch: 22 instructions instead of 51.
  ELL_HD static El add(const El& a, const El& b) {
    u32 t[17];
    (void)bn_add<17>(t, a.v, b.v);
    u32 c = t[16] >> 9;
    El r;
    u32 cc = 0;
    r.v[0] = addc32(t[0], c, cc, cc);
    ELL_UNROLL
    for (int i = 1; i < 16; i++) r.v[i] = t[i];
    r.v[16] = t[16] & 0x1FFu;
    if (ELL_UNLIKELY(cc != 0 || (r.v[16] == 0x1FFu && r.v[15] == 0xFFFFFFFFu))) {
      ELL_UNROLL
      for (int i = 1; i < 17; i++) r.v[i] = addc32(r.v[i], 0, cc, cc);
      // (a + b) - 2^521 + 1 <= 2^521 - 3: a folded sum cannot reach p; an unfolded one can equal it
      u32 ones = r.v[16] ^ 0x1FFu;
      ELL_UNROLL
      for (int i = 0; i < 16; i++) ones |= ~r.v[i];
      if (ones == 0) {
        ELL_UNROLL
        for (int i = 0; i < 17; i++) r.v[i] = 0;
      }
    }
    return r;
  }
  ELL_HD static El sub(const El& a, const El& b) {
    u32 t[17];
    u32 bw = bn_sub<17>(t, a.v, b.v);
    El r;
    u32 bb = 0;
    r.v[0] = subb32(t[0], bw, bb, bb);
    ELL_UNROLL
    for (int i = 1; i < 16; i++) r.v[i] = t[i];
    r.v[16] = t[16] + (bw << 9);
    if (ELL_UNLIKELY(bb != 0)) {
      ELL_UNROLL
      for (int i = 1; i < 16; i++) r.v[i] = subb32(t[i], 0, bb, bb);
      r.v[16] = t[16] + (bw << 9) - bb;
    }
    return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  // 2^K * a (K = 1..3): repeated doubling
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    El r = dbl(a);
    ELL_UNROLL
    for (int i = 1; i < K; i++) r = dbl(r);
    return r;
  }

  // value < 2^1088 in 34 limbs -> [0, p)
  ELL_HD static El reduce_wide(const u32 (&t)[34]) {
    u32 lo[17], hi[17];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) lo[i] = t[i];
    lo[16] = t[16] & 0x1FFu;
    ELL_UNROLL
    for (int i = 0; i < 17; i++) hi[i] = (t[16 + i] >> 9) | (i + 17 < 34 ? t[17 + i] << 23 : 0u);
    // hi may have up to 544 + ... bits when called from from_plain-sized inputs; for a product of
    // two residues it is < 2^521
    u32 r[17];
    u32 c = bn_add<17>(r, lo, hi);                 // < 2^522 (+ c for oversized inputs)
    // fold everything from bit 521 up once more
    u32 f = (r[16] >> 9) | (c << 23);
    r[16] &= 0x1FFu;
    u32 cc = 0;
    r[0] = addc32(r[0], f, cc, cc);
    ELL_UNROLL
    for (int i = 1; i < 17; i++) r[i] = addc32(r[i], 0, cc, cc);
    // now r <= 2^521: one more (rare) fold of bit 521, then r == p -> 0
    bool top = (r[16] >> 9) != 0;
    bool allones = true;
    ELL_UNROLL
    for (int i = 0; i < 16; i++) allones = allones && (r[i] == 0xFFFFFFFFu);
    allones = allones && (r[16] == 0x1FFu);
    if (ELL_UNLIKELY(top || allones)) {
      if (top) {                                   // r == 2^521  ->  1
        bn_zero<17>(r);
        r[0] = 1;
      } else {
        bn_zero<17>(r);
      }
    }
    El out;
    bn_copy<17>(out.v, r);
    return out;
  }
  ELL_HD static El from_plain(const u32 (&a)[17]) {           // a < 2^544
    u32 t[34];
    ELL_UNROLL
    for (int i = 0; i < 34; i++) t[i] = i < 17 ? a[i] : 0u;
    return reduce_wide(t);
  }
  ELL_HD static El mul_inline(const El& a, const El& b) {
    u32 t[34];
    fe_mul_wide<17>(t, a.v, b.v);
    return reduce_wide(t);
  }
  ELL_HD static El sqr_inline(const El& a) {
    u32 t[34];
    fe_sqr_wide<17>(t, a.v);
    return reduce_wide(t);
  }
  static ELL_HD_NOINLINE El mul_call(El a, El b) { return mul_inline(a, b); }
  static ELL_HD_NOINLINE El sqr_call(El a) { return sqr_inline(a); }
#if ELL_P521_INLINE
  ELL_HD static El mul(const El& a, const El& b) { return mul_inline(a, b); }
  ELL_HD static El sqr(const El& a) { return sqr_inline(a); }
#else
  ELL_HD static El mul(const El& a, const El& b) { return mul_call(a, b); }
  ELL_HD static El sqr(const El& a) { return sqr_call(a); }
#endif
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  static ELL_HD_NOINLINE El inv(const El& a) {
    El r;
    SafeGcd<consts::P521_P>::inv(r.v, a.v);
    return r;
  }
  // (p + 1) / 4 = 2^519: the square root is 519 squarings
  static ELL_HD_NOINLINE El sqrt(const El& a) { return sqr_n(a, 519); }
};

}  // namespace ell
