// ellgpu -- per-item work functions for short Weierstrass curves.
//
// Each function does the work of ONE GPU thread; the kernel template k_run<Fn>
// (hip_backend.h) calls them through the functors of engine.h with i = global thread
// id, tests/hostsim calls them in a loop on the CPU.  Memory layouts (all device buffers):
//
//   scalars / coordinates at the C ABI   big-endian, fixed width BYTES, item-major
//   jac   Jacobian results               SoA  jac[(c*L + limb)*n + i], c = X,Y,Z, field-internal form
//   tbl   per-item window tables         AoS  tbl[i*ENTRIES + e] (affine entries on the effective-
//                                             affine curve for secp256k1, Jacobian otherwise)
//   comb  fixed-base table               AoS  comb[w*(2^c - 1) + d-1]  (Aff structs, field-internal form)
//   pre   batch-inversion prefixes       SoA  pre[limb*n + i]
//   u12   ECDSA u1,u2                    SoA  u12[(c*LN + limb)*n + i], plain residues mod n
#pragma once

#include <type_traits>

#include "ladder.h"
#include "hmac_drbg.h"
#include "hmac_drbg256.h"
#include "hmac_drbg512.h"
#include "sha256.h"
#include "sha512.h"

// comb window width for the 256-bit curves (the CPU unit-test build of these headers
// overrides it with 8 so that it does not have to generate 2^20-entry tables on the host)
#ifndef ELL_COMB_BITS_256
#define ELL_COMB_BITS_256 16
#endif
// Short-curve comb of the 256-bit curves: SIGNED windows of this many bits (digits in
// [-2^(c-1), 2^(c-1)], table of the positive multiples, y negated at lookup).  22 bits: 12 windows
// x 2^21 affine entries = 1.6 GB per curve in use (of 288 GB), 12 mixed additions per k*G instead
// of the 16 of round 2's unsigned 16-bit comb (67 MB).  The CPU unit-test build (8 above) keeps
// its value.
#ifndef ELL_COMB_BITS_SHORT256
#if ELL_COMB_BITS_256 == 16
#define ELL_COMB_BITS_SHORT256 22
#else
#define ELL_COMB_BITS_SHORT256 ELL_COMB_BITS_256
#endif
#endif
#ifndef ELL_COMB_SIGNED_256
#define ELL_COMB_SIGNED_256 1
#endif
// 1 (default) = lambda*P entries are computed at lookup (x * beta) instead of being stored as a
// second window table: measured on one box (round 2, gpurun_out/r02d) ecdsa_main 8.61 -> 8.52 ms,
// FETCH_SIZE 6.03 -> 4.76 GB and WRITE_SIZE 2.20 -> 1.53 GB per 2^20 verifies (-24 % bytes)
#ifndef ELL_LAMBDA_AT_LOOKUP
#define ELL_LAMBDA_AT_LOOKUP 1
#endif

namespace ell {

template <class CV>
struct Work {
  typedef typename CV::F F;
  typedef typename CV::Fn Fn;
  typedef typename CV::C C;
  typedef typename F::El El;
  typedef typename Fn::El Nl;
  typedef ShortOps<CV> G;
  typedef Ladder<CV> LD;
  typedef Jac<F> J;
  typedef Aff<F> A;

  static constexpr int L = F::L;                           // 32-bit words of a plain value
  static constexpr int NS = sizeof(El) / sizeof(u32);      // stored limbs of a field element (9 for the 29-bit secp256k1 field)
  static constexpr int LN = Fn::L;
  static constexpr int BYTES = C::BYTES;
  static constexpr int NBYTES = C::NBYTES;
  static constexpr bool ENDO = CV::ENDO;
  // variable-base ladder geometry (see ladder.h)
  static constexpr int NSV = ENDO ? 2 : 1;                 // digit strings per (k, P)
  static constexpr int NNIB = ENDO ? 33 : 2 * BYTES;
  static constexpr bool TOP = !ENDO;
  static constexpr int NWIN = NNIB + (TOP ? 1 : 0);
  // variable-base window table: 8 affine odd multiples per digit string (16 entries per item:
  // P and lambda*P for secp256k1; the second half is build_table_odd8's scratch otherwise)
  // The secp256k1 odd-digit ladder (GLV halves below 2^130): WB-bit windows over the 2^(WB-1)
  // odd multiples.  4 bits = 33 windows (128 doublings, 65 additions) over 8 entries; 5 bits = 26
  // windows (125 doublings, 51 additions) over 16.  Since the table's entries cost a co-Z addition
  // and a rescaling each (1.7 k instructions; they were 2.6 k) the 5-bit form executes 3 % fewer
  // instructions -- and writes twice the table bytes: on a full grid (4 waves/SIMD) the two
  // measure the same (7.98 against 7.94 ms per 2^20), where only two or three waves per SIMD are
  // resident the 5-bit form is 2 % faster (profiles/r03_window_width_ab.txt).  So the small-grid
  // tuning (WIDE) takes 5 bits, the full-grid tuning 4.
  template <bool WIDE>
  struct Endo {
    static constexpr int WB = ENDO ? (WIDE ? ELL_ENDO_WBITS_WIDE : ELL_ENDO_WBITS) : 4;
    static constexpr int NW = (130 + WB - 1) / WB;                     // 26 (5 bits) / 33 (4 bits)
    static constexpr int NE = 1 << (WB - 1);                           // table entries
    static constexpr int TBL = 2 * NE > 16 ? 2 * NE : 16;             // slots per item: entries + the build's ratios
    static_assert(!ENDO || NW * 2 <= NWIN * NSV, "digit store too small for the odd ladder");
  };
  // the curves without an endomorphism: PLAIN_WB-bit odd windows over the full-width scalar
  // (8 BYTES bits + the forced top bit)
  static constexpr int PLAIN_WB = (L == 12) ? ELL_P384_WBITS : 4;
  static constexpr int PLAIN_NW = PLAIN_WB == 4 ? NWIN : (8 * BYTES + 1 + PLAIN_WB - 1) / PLAIN_WB;
  static constexpr int PLAIN_NE = 1 << (PLAIN_WB - 1);
  static_assert(ENDO || PLAIN_NW <= NWIN, "digit store too small");
  // slots between two items' tables in a launch of the given tuning
  template <bool WIDE>
  ELL_HD static constexpr int stride() {
    if constexpr (ENDO) return Endo<WIDE>::TBL;
    else return 2 * PLAIN_NE > 16 ? 2 * PLAIN_NE : 16;
  }
  // slots per item the scratch arena is sized for (the larger tuning's)
  static constexpr int TBL1 = ENDO ? (Endo<true>::TBL > Endo<false>::TBL ? Endo<true>::TBL : Endo<false>::TBL)
                                   : (2 * PLAIN_NE > 16 ? 2 * PLAIN_NE : 16);
  typedef A VT;
  static constexpr int TBLJ = 8 * NSV;                     // mul_add2: Jacobian entries per (k, P)
  // fixed-base comb: COMB_BITS-bit unsigned windows, table of d * 2^(COMB_BITS*w) * G.
  // 16-bit windows for the 256-bit curves (16 adds per k*G, 67 MB table that lives in
  // MALL/HBM and is gathered 64 B at a time); 8-bit windows otherwise.
  static constexpr int COMB_BITS = (L == 8) ? ELL_COMB_BITS_SHORT256 : 8;
  static constexpr bool COMB_SIGNED = (L == 8) && ELL_COMB_SIGNED_256;
  // signed windows: one more bit for the recoding's carry out of the top data window
  static constexpr int COMB_W = (8 * BYTES + (COMB_SIGNED ? 1 : 0) + COMB_BITS - 1) / COMB_BITS;
  static constexpr int COMB_DIG = COMB_SIGNED ? (1 << (COMB_BITS - 1)) : (1 << COMB_BITS) - 1;   // table entries per window
  static constexpr size_t COMB_ENTRIES = (size_t)COMB_W * COMB_DIG;

  // ---- I/O helpers -------------------------------------------------------
  ELL_HD static El load_fe(const u8* p) {
    u32 t[L];
    load_be<L>(t, p, BYTES);
    return F::from_plain(t);
  }
  ELL_HD static void store_fe(u8* p, const El& a) {
    u32 t[L];
    F::to_plain(t, a);
    store_be<L>(p, t, BYTES);
  }
  ELL_HD static A load_affine(const u8* xy, size_t i) {
    A a;
    a.x = load_fe(xy + i * 2 * BYTES);
    a.y = load_fe(xy + i * 2 * BYTES + BYTES);
    return a;
  }
  ELL_HD static void store_jac(u32* jac, size_t n, size_t i, const J& p) {
    ELL_UNROLL
    for (int l = 0; l < NS; l++) {
      jac[(size_t)(0 * NS + l) * n + i] = p.X.v[l];
      jac[(size_t)(1 * NS + l) * n + i] = p.Y.v[l];
      jac[(size_t)(2 * NS + l) * n + i] = p.Z.v[l];
    }
  }
  ELL_HD static J load_jac(const u32* jac, size_t n, size_t i) {
    J p;
    ELL_UNROLL
    for (int l = 0; l < NS; l++) {
      p.X.v[l] = jac[(size_t)(0 * NS + l) * n + i];
      p.Y.v[l] = jac[(size_t)(1 * NS + l) * n + i];
      p.Z.v[l] = jac[(size_t)(2 * NS + l) * n + i];
    }
    return p;
  }

  // beta (curves.js:189-198) in the field's internal form
  ELL_HD static El load_beta() {
    if constexpr (ENDO) {
      u32 b[L];
      ELL_UNROLL
      for (int i = 0; i < L; i++) b[i] = C::beta[i];
      return F::from_plain(b);
    } else {
      return F::zero();
    }
  }

  // ---- variable base -----------------------------------------------------
  // Recode k and build the window table(s) of P into digit slots
  // [s0, s0+NSV) of an NS-string store and table entries tbl[0 .. TBLJ).
  ELL_HD static void prepare_var(const u32 (&k)[L], const A& p, const DigitStore& ds, int s0,
                                 int NS, J* tbl, u32& negmask) {
    LD::build_table8(tbl, p);
    if constexpr (ENDO) {
      u32 k1[5], k2[5];
      bool n1, n2;
      glv_split(k, k1, n1, k2, n2);
      recode_w4<5, NNIB, false>(k1, ds, s0, NS);
      recode_w4<5, NNIB, false>(k2, ds, s0 + 1, NS);
      negmask |= (n1 ? 1u : 0u) << s0;
      negmask |= (n2 ? 1u : 0u) << (s0 + 1);
      // lambda*P table: (beta*X, Y, Z)   (short.js:282-310 _getBeta)
      El beta = load_beta();
      ELL_NOUNROLL
      for (int e = 0; e < 8; e++) {
        J t = tbl[e];
        t.X = F::mul(t.X, beta);
        tbl[8 + e] = t;
      }
    } else {
      recode_w4<L, NNIB, true>(k, ds, s0, NS);
    }
  }

  // k*P for one (k, P), result in true Jacobian coordinates.  secp256k1: GLV split, odd
  // signed digits, effective-affine tables of P and lambda*P (mixed adds only); other
  // curves: odd signed digits over an affine table of the odd multiples.
  // WIDE = the register-rich tuning of the secp256k1 ladder (common.h): beta and zg stay in
  // registers, table entries are requested one step ahead (Ladder::run_odd_w4)
  template <bool WIDE = false>
  ELL_HD static J var_ladder(const u32 (&k)[L], const A& p, VT* tbl, const DigitStore& ds, bool& inf) {
    if constexpr (ENDO) {
      typedef Endo<WIDE> E;
      u32 k1[5], k2[5];
      bool n1, n2;
      glv_split<true>(k, k1, n1, k2, n2);           // both halves odd: no correction at the end
      const u32 evenmask = 0u;
      recode_odd_w4<5, E::NW, E::WB>(k1, ds, 0, 2);
      recode_odd_w4<5, E::NW, E::WB>(k2, ds, 1, 2);
      u32 negmask = (n1 ? 1u : 0u) | (n2 ? 2u : 0u);
      El zg;
      LD::template build_table_odd8<E::NE>(tbl, p, zg);
      // lambda*P table: (beta*x, y)   (short.js:282-310 _getBeta); beta commutes with the
      // isomorphisms, which only scale x and y
      El beta = load_beta();
#if ELL_LAMBDA_AT_LOOKUP
      J r;
#if ELL_SPILL_ZG && defined(__HIP_DEVICE_COMPILE__)
      if constexpr (!WIDE) {
        // zg is needed again only after the ladder: park it in a free table slot (slots 8..15 are
        // the build's scratch) instead of eight registers held across the loop
        tbl[2 * E::NE - 1].x = zg;
        r = LD::template run_odd_w4<2, E::NW, true, false, E::WB>(ds, tbl, negmask, evenmask, inf, &beta);
        asm volatile("" ::: "memory");
        zg = tbl[2 * E::NE - 1].x;
      } else
#endif
      {
        r = LD::template run_odd_w4<2, E::NW, true, WIDE, E::WB>(ds, tbl, negmask, evenmask, inf, &beta);
      }
#else
      ELL_NOUNROLL
      for (int e = 0; e < E::NE; e++) {
        A t = tbl[e];
        t.x = F::mul(t.x, beta);
        tbl[E::NE + e] = t;
      }
      J r = LD::template run_odd_w4<2, E::NW, false, false, E::WB>(ds, tbl, negmask, evenmask, inf);
#endif
      r.Z = F::mul(r.Z, zg);
      return r;
    } else if constexpr ((L > 12 && ELL_P521_JTABLE) || CV::JTABLE) {
      // (developer switch, see fp.h) plain signed-window ladder over a Jacobian table: the
      // 16 affine slots hold its 8 Jacobian entries
      static_assert(16 * sizeof(A) >= 8 * sizeof(J), "table slot too small");
      u32 negmask = 0;
      prepare_var(k, p, ds, 0, NSV, (J*)tbl, negmask);
      J r = LD::template run_w4<NSV, NWIN>(ds, (const J*)tbl, negmask);
      inf = G::is_inf(r);
      return r;
    } else {
      // no endomorphism: the same odd-digit ladder over the full-width scalar.  The table is
      // built on the isomorphic curves (additions do not involve a) and mapped back to the
      // true curve -- whose a = -3 doubling the ladder needs -- by one division-step inversion
      // of the common factor zg: x = x' zg^-2, y = y' zg^-3.  Mixed additions throughout.
      u32 kk[L];
      bn_copy<L>(kk, k);
      u32 evenmask = (k[0] & 1u) ? 0u : 1u;
      kk[0] |= 1u;                                  // k even -> k + 1, P subtracted at the end
      recode_odd_w4<L, PLAIN_NW, PLAIN_WB>(kk, ds, 0, 1);
      El zg;
      LD::template build_table_odd8<PLAIN_NE>(tbl, p, zg);
      El zi = F::inv(zg);
      El zi2 = F::sqr(zi);
      El zi3 = F::mul(zi2, zi);
      ELL_NOUNROLL
      for (int e = 0; e < PLAIN_NE; e++) {
        A t = tbl[e];
        t.x = F::mul(t.x, zi2);
        t.y = F::mul(t.y, zi3);
        tbl[e] = t;
      }
      return LD::template run_odd_w4<1, PLAIN_NW, false, false, PLAIN_WB>(ds, tbl, 0u, evenmask, inf);
    }
  }

  // k*P -> Jacobian (Point#mul's ladder: short.js:422-432 -> base.js:86-126 /
  // short.js:218-249)
  template <bool WIDE = false>
  ELL_HD static void mul_var(size_t i, size_t n, const u8* ks, const u8* xy, VT* tbl_all,
                             const DigitStore& ds, u32* jac) {
    u32 k[L];
    load_be<L>(k, ks + i * BYTES, BYTES);
    A p = load_affine(xy, i);
    bool inf;
    J r = var_ladder<WIDE>(k, p, tbl_all + i * stride<WIDE>(), ds, inf);
    store_jac(jac, n, i, r);
  }

  // k1*P1 + k2*P2 -> Jacobian, both points per-item (mulAdd/jmulAdd,
  // short.js:434-450 -> base.js:128-253): one ladder, shared doublings
  ELL_HD static void mul_add2(size_t i, size_t n, const u8* k1s, const u8* xy1, const u8* k2s,
                              const u8* xy2, J* tbl_all, const DigitStore& ds, u32* jac) {
    u32 k1[L], k2[L];
    load_be<L>(k1, k1s + i * BYTES, BYTES);
    load_be<L>(k2, k2s + i * BYTES, BYTES);
    A p1 = load_affine(xy1, i);
    A p2 = load_affine(xy2, i);
    J* tbl = tbl_all + i * 2 * TBLJ;
    u32 negmask = 0;
    prepare_var(k1, p1, ds, 0, 2 * NSV, tbl, negmask);
    prepare_var(k2, p2, ds, NSV, 2 * NSV, tbl + TBLJ, negmask);
    J r = LD::template run_w4<2 * NSV, NWIN>(ds, tbl, negmask);
    store_jac(jac, n, i, r);
  }

  // k1*G + k2*P2 -> Jacobian (the shape ECDSA verify uses): window ladder for P2, then the
  // comb's mixed additions for G onto the same accumulator (no second point, no Jacobian add)
  ELL_HD static void mul_add_g_item(size_t i, size_t n, const u8* k1s, const u8* k2s,
                                    const u8* xy2, const A* comb, VT* tbl_all,
                                    const DigitStore& ds, u32* jac) {
    u32 k1[L], k2[L];
    load_be<L>(k2, k2s + i * BYTES, BYTES);
    A p2 = load_affine(xy2, i);
    bool inf;
    J b = var_ladder(k2, p2, tbl_all + i * stride<false>(), ds, inf);
#if ELL_LATE_LOADS && defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");                 // k1 is loaded after the ladder (see ecdsa_main)
#endif
    load_be<L>(k1, k1s + i * BYTES, BYTES);
    J r = LD::template comb_add<L, COMB_W, COMB_BITS, false, COMB_SIGNED>(b, inf, k1, comb);
    store_jac(jac, n, i, r);
  }

  // ---- affine point addition ---------------------------------------------------
  // Point#add (short.js:365-392, with #dbl :394-412 for equal points): P + Q for affine
  // inputs, O on either side, P == Q and P == -Q included -> Jacobian (normalized by the
  // caller's batch inversion, where the reference inverts per addition).  The chord and
  // tangent formulas are the reference's own, so off-curve inputs give its results too.
  ELL_HD static void point_add(size_t i, size_t n, const u8* xy1, const u8* inf1, const u8* xy2,
                               const u8* inf2, u32* jac) {
    A p = load_affine(xy1, i);
    A q = load_affine(xy2, i);
    bool pinf = inf1 && inf1[i];
    bool qinf = inf2 && inf2[i];
    J P = G::select(pinf, G::infinity(), G::from_affine(p));
    J r = G::add_mixed(P, q);
    r = G::select(qinf, P, r);                      // P + O = P, O + O = O
    store_jac(jac, n, i, r);
  }

  // ---- fixed base ----------------------------------------------------------
  // k*G -> Jacobian (replaces _fixedNafMul, base.js:52-84)
  ELL_HD static void mul_fixed(size_t i, size_t n, const u8* ks, const A* comb, u32* jac) {
    u32 k[L];
    load_be<L>(k, ks + i * BYTES, BYTES);
    J r = LD::template comb_mul<L, COMB_W, COMB_BITS, COMB_SIGNED>(k, comb);
    store_jac(jac, n, i, r);
  }

  // ---- point decompression (ShortCurve#pointFromX, short.js:187-204) --------------
  // rhs = x^3 + a x + b  (ShortCurve#validate short.js:205-216 and pointFromX share it)
  ELL_HD static El curve_rhs(const El& x) {
    El x2 = F::sqr(x);
    if constexpr (CV::A_KIND == 1) {
      // user-defined curve: a and b come from the run-time parameter block (fp_rt.h)
      return F::add(F::mul(F::add(x2, F::curve_a()), x), F::curve_b());
    } else {
    u32 bp[L];
    ELL_UNROLL
    for (int l = 0; l < L; l++) bp[l] = C::b_plain[l];
    El b = F::from_plain(bp);
    if (CV::A_KIND == 0) return F::add(F::mul(x2, x), b);
    El three = F::add(F::one(), F::dbl(F::one()));
    return F::add(F::mul(F::sub(x2, three), x), b);            // x^3 - 3x + b
    }
  }
  // ---- the engine's domain: points ON the curve -----------------------------------------
  // The reference never validates a point on this path (ec/index.js:192 keyFromPublic,
  // ec/key.js:27-35, short.js:422-432): it runs its formulas on any (x, y).  Off the curve
  // those formulas are no group law -- the result depends on the exact order of the
  // reference's operations (its wNAF / JSF digits, its GLV split, the window of G's shipped
  // table), which the ladders here deliberately do not share.  So an off-curve operand is
  // OUTSIDE the engine's domain: it is detected (one squaring more than ShortCurve#validate,
  // short.js:205-216) and reported per item -- out_inf = 2, or out_status = 2 beside a verdict
  // of 0, at the C ABI -- never answered with a guess; the JS layer hands such items to the
  // reference's own method.
  enum { DOMAIN_OFF_CURVE = 2 };
  ELL_HD static bool on_curve(const A& a) { return F::eq(F::sqr(a.y), curve_rhs(a.x)); }
  // EC#verify's answer for item i.  out_ok is a MASK, strictly 0 / 1: anything the engine cannot
  // vouch for is 0 there, so a caller that reads it as a boolean never accepts a signature over a
  // key that is no curve point.  The domain status goes to out_st (may be null): 2 where r and s
  // are in range but the key is not on the curve (the reference computes with such a key, and can
  // even answer true -- a caller that wants ITS answer runs it on those items), else 0.
  ELL_HD static void store_verdict(size_t i, u8 valid, bool on, bool ok, u8* out_ok, u8* out_st) {
    const bool in_range = valid != 0;
    out_ok[i] = (in_range && on && ok) ? (u8)1 : (u8)0;
    if (out_st) out_st[i] = (in_range && !on) ? (u8)DOMAIN_OFF_CURVE : (u8)0;
  }
  // after the ladder + normalization of a point-valued call: items with an operand that is not
  // on the curve get out_inf = 2 and a zeroed result (xy1 / xy2: the call's point operands,
  // either may be null)
  ELL_HD static void domain_mark(size_t i, const u8* xy1, const u8* xy2, u8* out_xy, u8* out_inf) {
    bool on = true;
    if (xy1) on = on_curve(load_affine(xy1, i));
    if (xy2) on = on && on_curve(load_affine(xy2, i));
    if (on) return;
    out_inf[i] = (u8)DOMAIN_OFF_CURVE;
    if (out_xy) {
      ELL_NOUNROLL
      for (int b = 0; b < 2 * BYTES; b++) out_xy[i * 2 * BYTES + b] = 0;
    }
  }
  // y = sqrt(x^3 + a x + b) with the requested parity; false ('invalid point') when x is
  // not the abscissa of a curve point.  One exponentiation.
  ELL_HD static bool lift_x(El& y, const El& x, bool want_odd) {
    El rhs = curve_rhs(x);
    y = F::sqrt(rhs);
    bool ok = F::eq(F::sqr(y), rhs);
    u32 yp[L];
    F::to_plain(yp, y);
    bool is_odd = (yp[0] & 1u) != 0;
    El yn = F::neg(y);
    y = fe_select<F>(is_odd != want_odd, yn, y);
    return ok;
  }
  ELL_HD static void decompress(size_t i, const u8* xs, const u8* odd, u8* out_xy, u8* out_ok) {
    El x = load_fe(xs + i * BYTES);
    El y;
    bool ok = lift_x(y, x, odd[i] != 0);
    if (!ok) { x = F::zero(); y = F::zero(); }
    store_fe(out_xy + i * 2 * BYTES, x);
    store_fe(out_xy + i * 2 * BYTES + BYTES, y);
    out_ok[i] = ok ? 1 : 0;
  }

  // ---- SEC1 codecs --------------------------------------------------------------
  // BaseCurve#decodePoint (base.js:270-293) of n encodings of `len` bytes each.  status:
  // 0 = point, 1 = 'Unknown point format' (prefix / length), 2 = 'invalid point' (compressed x
  // without a y), 3 = 'Assertion failed' (hybrid 06/07 prefix contradicting y's last bit).
  // Like the reference, an uncompressed encoding is NOT checked against the curve equation and
  // its coordinates are reduced mod p (Point's toRed, short.js:261-264).
  enum { DECODE_OK = 0, DECODE_FORMAT = 1, DECODE_INVALID = 2, DECODE_ASSERT = 3 };
  ELL_HD static void decode_point(size_t i, const u8* enc, size_t len, u8* out_xy, u8* status) {
    const u8* e = enc + i * len;
    const u32 tag = len ? e[0] : 0u;
    El x = F::zero(), y = F::zero();
    u32 st = DECODE_FORMAT;
    if ((tag == 4 || tag == 6 || tag == 7) && len == 1 + 2 * (size_t)BYTES) {
      const u32 last = e[len - 1] & 1u;
      if ((tag == 6 && last != 0) || (tag == 7 && last != 1)) st = DECODE_ASSERT;
      else {
        x = load_fe(e + 1);
        y = load_fe(e + 1 + BYTES);
        st = DECODE_OK;
      }
    } else if ((tag == 2 || tag == 3) && len == 1 + (size_t)BYTES) {
      if constexpr (F::HAS_SQRT) {
        x = load_fe(e + 1);
        // no y for this x: 'invalid point' (short.js:196-197) -- except over p224, where bn.js's
        // Tonelli-Shanks loop gives up first ('Assertion failed', dist/elliptic.js:7296)
        st = lift_x(y, x, tag == 3) ? DECODE_OK : (CV::ID == CURVE_P224 ? DECODE_ASSERT : DECODE_INVALID);
        if (st != DECODE_OK) { x = F::zero(); y = F::zero(); }
      }
    }
    store_fe(out_xy + i * 2 * BYTES, x);
    store_fe(out_xy + i * 2 * BYTES + BYTES, y);
    status[i] = (u8)st;
  }
  // BasePoint#_encode (base.js:299-307): 02/03 || x  or  04 || x || y, coordinates reduced
  ELL_HD static void encode_point(size_t i, const u8* xy, int compact, u8* out) {
    A a = load_affine(xy, i);
    const size_t len = compact ? 1 + (size_t)BYTES : 1 + 2 * (size_t)BYTES;
    u8* o = out + i * len;
    u32 yp[L];
    F::to_plain(yp, a.y);
    o[0] = compact ? (u8)(2u + (yp[0] & 1u)) : (u8)4;
    store_fe(o + 1, a.x);
    if (!compact) store_fe(o + 1 + BYTES, a.y);
  }
  // ---- signature DER codec (lib/elliptic/ec/signature.js) ---------------------------
  // getLength (signature.js:30-59).  false = the reference's `false`; a length byte read past
  // the end (JavaScript `undefined`) also ends in `return false` at the caller's next test.
  ELL_HD static bool der_length(const u8* d, u64 dl, u64& place, u64& out) {
    if (place >= dl) return false;
    u32 initial = d[place++];
    if (!(initial & 0x80u)) { out = initial; return true; }
    u32 oct = initial & 0xFu;
    if (oct == 0 || oct > 4) return false;                    // indefinite length or overflow
    if (place < dl && d[place] == 0) return false;
    u64 val = 0, off = place;
    ELL_NOUNROLL
    for (u32 i = 0; i < oct; i++, off++) val = ((val << 8) | (off < dl ? (u64)d[off] : 0ull)) & 0xFFFFFFFFull;
    if (val <= 0x7F) return false;                            // leading zeroes
    place = off;
    out = val;
    return true;
  }
  // Signature#_importDER (signature.js:83-147) of d[0..dl).  status 0: r and s written as
  // NBYTES-wide big-endian integers; 1: the reference's `return false` ("Signature without r
  // or s" is thrown by the constructor); 2: well-formed, but r or s is wider than NBYTES (hence
  // >= n: EC#verify answers false) -- r and s are zeroed for 1 and 2.
  enum { DER_OK = 0, DER_MALFORMED = 1, DER_TOO_WIDE = 2 };
  ELL_HD static u32 der_parse(const u8* d, u64 dl, u64& r0, u64& rn, u64& s0, u64& sn) {
    u64 p = 0;
    if (!(p < dl && d[p] == 0x30)) return DER_MALFORMED;
    p++;
    u64 len;
    if (!der_length(d, dl, p, len)) return DER_MALFORMED;
    if (len + p != dl) return DER_MALFORMED;
    if (!(p < dl && d[p] == 0x02)) return DER_MALFORMED;
    p++;
    u64 rlen;
    if (!der_length(d, dl, p, rlen)) return DER_MALFORMED;
    if (p < dl && (d[p] & 0x80u)) return DER_MALFORMED;
    r0 = p;
    p += rlen;
    if (!(p < dl && d[p] == 0x02)) return DER_MALFORMED;       // also: r lies inside the buffer
    p++;
    u64 slen;
    if (!der_length(d, dl, p, slen)) return DER_MALFORMED;
    if (dl != slen + p) return DER_MALFORMED;
    if (p < dl && (d[p] & 0x80u)) return DER_MALFORMED;
    s0 = p;
    rn = rlen;
    sn = slen;
    if (rn > 0 && d[r0] == 0) {
      if (rn > 1 && (d[r0 + 1] & 0x80u)) { r0++; rn--; } else return DER_MALFORMED;
    }
    if (sn > 0 && d[s0] == 0) {
      if (sn > 1 && (d[s0 + 1] & 0x80u)) { s0++; sn--; } else return DER_MALFORMED;
    }
    return (rn > (u64)NBYTES || sn > (u64)NBYTES) ? DER_TOO_WIDE : DER_OK;
  }
  ELL_HD static void sig_from_der(size_t i, const u8* der, size_t stride, const u32* der_len,
                                  u8* out_r, u8* out_s, u8* status) {
    const u8* d = der + i * stride;
    u64 dl = der_len[i];
    u64 r0 = 0, rn = 0, s0 = 0, sn = 0;
    u32 st = dl > stride ? (u32)DER_MALFORMED : der_parse(d, dl, r0, rn, s0, sn);
    u8* ro = out_r + i * NBYTES;
    u8* so = out_s + i * NBYTES;
    ELL_NOUNROLL
    for (int j = 0; j < NBYTES; j++) {
      u64 back = (u64)(NBYTES - j);                            // distance from the end
      ro[j] = (st == DER_OK && back <= rn) ? d[r0 + rn - back] : (u8)0;
      so[j] = (st == DER_OK && back <= sn) ? d[s0 + sn - back] : (u8)0;
    }
    status[i] = (u8)st;
  }
  // Signature#toDER (signature.js:149-176) of (r, s) given NBYTES wide; out_len[i] = bytes
  // written at out + i * stride (stride >= 2 * NBYTES + 9; the rest of the record is zeroed).  s = 0 sends the reference into an
  // endless loop (signature.js:162-164): reported as length 0.
  ELL_HD static void sig_to_der(size_t i, const u8* r, const u8* s, u8* out, size_t stride, u32* out_len) {
    const u8* rp = r + i * NBYTES;
    const u8* sp = s + i * NBYTES;
    u8* o = out + i * stride;
    int rz = 0, sz = 0;                                        // leading zero bytes; BN#toArray() of 0 is [0]
    while (rz < NBYTES - 1 && rp[rz] == 0) rz++;
    while (sz < NBYTES && sp[sz] == 0) sz++;
    if (sz == NBYTES) {
      ELL_NOUNROLL
      for (size_t j = 0; j < stride; j++) o[j] = 0;
      out_len[i] = 0;
      return;
    }
    int rn = NBYTES - rz, sn = NBYTES - sz;
    int rpad = (rp[rz] & 0x80u) ? 1 : 0, spad = (sp[sz] & 0x80u) ? 1 : 0;
    int body = 2 + rpad + rn + 2 + spad + sn;                  // both integer lengths are < 0x80
    int pos = 0;
    o[pos++] = 0x30;
    if (body >= 0x80) o[pos++] = 0x81;                         // constructLength (signature.js:136-147)
    o[pos++] = (u8)body;
    o[pos++] = 0x02;
    o[pos++] = (u8)(rpad + rn);
    if (rpad) o[pos++] = 0;
    ELL_NOUNROLL
    for (int j = 0; j < rn; j++) o[pos++] = rp[rz + j];
    o[pos++] = 0x02;
    o[pos++] = (u8)(spad + sn);
    if (spad) o[pos++] = 0;
    ELL_NOUNROLL
    for (int j = 0; j < sn; j++) o[pos++] = sp[sz + j];
    out_len[i] = (u32)pos;
    ELL_NOUNROLL
    for (size_t j = (size_t)pos; j < stride; j++) o[j] = 0;     // the rest of the record is defined
  }
  // EC#verify on wire formats: exceptions in the reference's order (keyFromPublic first, then
  // the Signature constructor); err 1..3 = decodePoint's status, 4 = 'Signature without r or s'
  // err 5 (ok = 0 like every other err): an uncompressed key that is not on the curve -- no
  // exception of the reference's, it computes with such keys; outside the engine's domain
  // (on_curve below; ver_st = the verify's status array, Work::store_verdict)
  ELL_HD static void wire_status(size_t i, const u8* key_st, const u8* sig_st, const u8* ver_st, u8* ok, u8* err) {
    u32 e = key_st[i] ? key_st[i] : (sig_st[i] == DER_MALFORMED ? 4u : 0u);
    if (e || sig_st[i] == DER_TOO_WIDE) ok[i] = 0;
    else if (ver_st[i] == (u8)DOMAIN_OFF_CURVE) e = 5u;
    if (err) err[i] = (u8)e;
  }

  // KeyPair#validate (ec/key.js:41-52), first two tests: 1 = 'Invalid public key' (infinity),
  // 2 = 'Public key is not a point' (ShortCurve#validate short.js:205-216), else 0
  enum { VALIDATE_OK = 0, VALIDATE_INF = 1, VALIDATE_NOT_POINT = 2, VALIDATE_ORDER = 3 };
  ELL_HD static void validate_point(size_t i, const u8* xy, const u8* inf, u8* status) {
    if (inf && inf[i]) { status[i] = VALIDATE_INF; return; }
    A a = load_affine(xy, i);
    status[i] = F::eq(F::sqr(a.y), curve_rhs(a.x)) ? VALIDATE_OK : VALIDATE_NOT_POINT;
  }
  // the group order as a scalar for item i (third test: pub.mul(n).isInfinity())
  ELL_HD static void fill_order(size_t i, u8* scal) {
    u32 nq[LN], nn[L];
    Fn::get_p(nq);
    ELL_UNROLL
    for (int l = 0; l < L; l++) nn[l] = l < LN ? nq[l] : 0u;
    store_be<L>(scal + i * BYTES, nn, BYTES);
  }

  // ---- Jacobian -> affine with Montgomery's trick ------------------------------
  // Thread t converts items t, t+T, t+2T, ... (< n), K of them, with ONE field
  // inversion (replaces the per-point redInvm of JPoint#toP, short.js:516-526).
  // out_xy: big-endian x||y (zeroed for infinity), out_inf: 1 = infinity.
  // raw_aff != null: additionally store the affine point in field-internal
  // form (used to build the comb tables).
  ELL_HD static void normalize(size_t t, size_t T, size_t n, int K, const u32* jac, u32* pre,
                               u8* out_xy, u8* out_inf, A* raw_aff) {
    // Both passes are software-pipelined by hand (round 4): item j + 1's words are requested
    // before item j's multiplications, so that the thread's chain -- K items and one inversion,
    // on a grid of n / K threads that leaves most SIMDs a single wave -- does not also wait for
    // HBM once per item.
    auto load_z = [&](size_t i) {
      El z;
      ELL_UNROLL
      for (int l = 0; l < NS; l++) z.v[l] = jac[(size_t)(2 * NS + l) * n + i];
      return z;
    };
    // items of this thread: t, t + T, ... below n
    int cnt = 0;
    if (t < n) cnt = (int)((n - 1 - t) / T) + 1;
    if (cnt > K) cnt = K;
    El acc = F::one();
    El znext = cnt > 0 ? load_z(t) : F::one();
    ELL_NOUNROLL
    for (int j = 0; j < cnt; j++) {
      size_t i = t + (size_t)j * T;
      El z = znext;
      if (j + 1 < cnt) znext = load_z(i + T);
      bool inf = F::is_zero(z);
      z = fe_select<F>(inf, F::one(), z);
      ELL_UNROLL
      for (int l = 0; l < NS; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = F::mul(acc, z);
    }
    El inv = F::inv(acc);
    J pn;
    El prn;
    auto load_item = [&](size_t i, J& p, El& pr) {
      p = load_jac(jac, n, i);
      ELL_UNROLL
      for (int l = 0; l < NS; l++) pr.v[l] = pre[(size_t)l * n + i];
    };
    if (cnt > 0) load_item(t + (size_t)(cnt - 1) * T, pn, prn);
    ELL_NOUNROLL
    for (int j = cnt - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      J p = pn;
      El pr = prn;
      if (j > 0) load_item(i - T, pn, prn);
      bool inf = F::is_zero(p.Z);
      El z = fe_select<F>(inf, F::one(), p.Z);
      El zinv = F::mul(inv, pr);
      inv = F::mul(inv, z);
      El zi2 = F::sqr(zinv);
      El x = F::mul(p.X, zi2);
      El y = F::mul(p.Y, F::mul(zi2, zinv));
      if (inf) { x = F::zero(); y = F::zero(); }
      if (out_xy) {
        store_fe(out_xy + i * 2 * BYTES, x);
        store_fe(out_xy + i * 2 * BYTES + BYTES, y);
      }
      if (out_inf) out_inf[i] = inf ? 1 : 0;
      if (raw_aff) { raw_aff[i].x = x; raw_aff[i].y = y; }
    }
  }

  // ---- ECDSA verify (ec/index.js:188-229) ---------------------------------------
  // z = H >> shift for the hash_len-byte big-endian H  (_truncateToN,
  // ec/index.js:81-108; shift = max(0, msgBits - n.bitLength()))
  ELL_HD static void load_hash(u32 (&e)[LN], const u8* h, int hash_len, int shift) {
    int hb = hash_len - (shift >> 3);
    int bs = shift & 7;
    u32 t[LN + 1];
    if (hb <= 0) {
      bn_zero<LN>(e);
      return;
    }
    load_be<LN + 1>(t, h, hb);
    ELL_UNROLL
    for (int i = 0; i < LN; i++) e[i] = bs ? ((t[i] >> bs) | (t[i + 1] << (32 - bs))) : t[i];
  }

  ELL_HD static bool scalar_in_range(const u32 (&x)[LN]) {      // 1 <= x < n
    u32 nn[LN];
    ELL_UNROLL
    for (int i = 0; i < LN; i++) nn[i] = C::n[i];
    return !bn_is_zero<LN>(x) && !bn_geq<LN>(x, nn);
  }

  // Pass 1: thread t handles items t, t+T, ... (K of them): range checks,
  // s^-1 mod n for all K with one inversion, u1 = z/s, u2 = r/s.
  ELL_HD static void ecdsa_prep(size_t t, size_t T, size_t n, int K, const u8* hash,
                                int hash_len, int shift, const u8* rs, const u8* ss, u32* pre,
                                u32* u12, u8* valid) {
    Nl acc = Fn::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      u32 r[LN], s[LN];
      load_be<LN>(r, rs + i * NBYTES, NBYTES);
      load_be<LN>(s, ss + i * NBYTES, NBYTES);
      bool ok = scalar_in_range(r) && scalar_in_range(s);
      valid[i] = ok ? 1 : 0;
      Nl sm = Fn::from_plain(s);
      sm = fe_select<Fn>(ok, sm, Fn::one());
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = Fn::mul(acc, sm);
    }
    Nl inv = Fn::inv(acc);
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      u32 r[LN], s[LN], e[LN];
      load_be<LN>(r, rs + i * NBYTES, NBYTES);
      load_be<LN>(s, ss + i * NBYTES, NBYTES);
      load_hash(e, hash + i * (size_t)hash_len, hash_len, shift);
      bool ok = valid[i] != 0;
      Nl sm = fe_select<Fn>(ok, Fn::from_plain(s), Fn::one());
      Nl pr;
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pr.v[l] = pre[(size_t)l * n + i];
      Nl w = Fn::mul(inv, pr);                       // s^-1 (Montgomery form: s^-1 R)
      inv = Fn::mul(inv, sm);
      // u = x * s^-1 as a PLAIN residue in one Montgomery product: (x) * (s^-1 R) / R, x < R
      // unreduced on the left, s^-1 R < n on the right -- no conversion of x in, none of u out
      // (round 3: 7 instead of 11 products per item)
      Nl er, rr;
      bn_copy<LN>(er.v, e);
      bn_copy<LN>(rr.v, r);
      Nl u1 = Fn::mul(er, w);
      Nl u2 = Fn::mul(rr, w);
      u32 p1[LN], p2[LN];
      bn_copy<LN>(p1, u1.v);
      bn_copy<LN>(p2, u2.v);
      ELL_UNROLL
      for (int l = 0; l < LN; l++) {
        u12[(size_t)(0 * LN + l) * n + i] = ok ? p1[l] : 0u;
        u12[(size_t)(1 * LN + l) * n + i] = ok ? p2[l] : 0u;
      }
    }
  }

  // ---- public-key recovery (ec/index.js:231-259 EC#recoverPubKey) ----------------------
  // Q = r^-1 (s R - e G) with R = pointFromX(r + (j >> 1) n, j & 1); e = new BN(msg) is NOT
  // truncated there, only reduced by the arithmetic mod n.
  // Pass 1 (thread t handles items t, t+T, ...: one inversion per K items): checks, the
  // x-coordinate and parity for the decompression kernel, s1 = -e/r and s2 = s/r for the
  // double-scalar kernel.  status: 0 so far fine, 2 the reference throws (second candidate
  // with r >= p mod n, j > 3), 3 outside the engine's domain (r = 0 or r >= n: the reference
  // goes through BN#invm of an unreduced value there; callers hand those to the reference).
  enum { RECOVER_POINT = 0, RECOVER_INF = 1, RECOVER_THROWS = 2, RECOVER_DOMAIN = 3 };
  // a big-endian byte string of up to 8 LN bytes -> its value mod n, in Montgomery form
  ELL_HD static Nl bytes_mod_n(const u8* h, int len) {
    u32 t[2 * LN];
    load_be<2 * LN>(t, h, len);
    u32 lo[LN], hi[LN];
    ELL_UNROLL
    for (int i = 0; i < LN; i++) { lo[i] = t[i]; hi[i] = t[LN + i]; }
    // hi * 2^(32 LN) + lo: from_plain(hi) = hi * R as a residue, read as a plain value
    Nl him = Fn::from_plain(hi);
    return Fn::add(Fn::from_plain(lo), Fn::from_plain(him.v));
  }
  ELL_HD static void recover_prep(size_t t, size_t T, size_t n, int K, const u8* hash, int hash_len,
                                  const u8* rs, const u8* ss, const u8* recid, u32* pre, u8* xs,
                                  u8* odd, u8* s1b, u8* s2b, u8* status) {
    static_assert(L == LN, "the presets' fields and orders have the same limb count");
    u32 nn[LN], pmn[LN];
    {
      u32 pp[L];
      F::get_p(pp);
      ELL_UNROLL
      for (int i = 0; i < LN; i++) nn[i] = C::n[i];
      bn_sub<LN>(pmn, pp, nn);                   // p mod n = p - n  (n < p < 2n for every preset)
    }
    Nl acc = Fn::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      u32 r[LN];
      load_be<LN>(r, rs + i * NBYTES, NBYTES);
      u32 jj = recid[i];
      bool second = (jj >> 1) != 0;
      int st = RECOVER_POINT;
      if (!scalar_in_range(r)) st = RECOVER_DOMAIN;
      else if (jj > 3 || (second && bn_geq<LN>(r, pmn))) st = RECOVER_THROWS;
      status[i] = (u8)st;
      u32 x[LN], rn[LN];
      bn_add<LN>(rn, r, nn);                     // r + n < p when the second candidate exists
      bn_select<LN>(x, second && st == RECOVER_POINT, rn, r);
      store_be<LN>(xs + i * BYTES, x, BYTES);
      odd[i] = (u8)(jj & 1);
      Nl rm = fe_select<Fn>(st == RECOVER_POINT, Fn::from_plain(r), Fn::one());
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = Fn::mul(acc, rm);
    }
    Nl inv = Fn::inv(acc);
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      u32 r[LN], s[LN];
      load_be<LN>(r, rs + i * NBYTES, NBYTES);
      load_be<LN>(s, ss + i * NBYTES, NBYTES);
      bool ok = status[i] == RECOVER_POINT;
      Nl rm = fe_select<Fn>(ok, Fn::from_plain(r), Fn::one());
      Nl pr;
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pr.v[l] = pre[(size_t)l * n + i];
      Nl rinv = Fn::mul(inv, pr);
      inv = Fn::mul(inv, rm);
      Nl e = bytes_mod_n(hash + i * (size_t)hash_len, hash_len);
      Nl s1 = Fn::neg(Fn::mul(e, rinv));
      Nl s2 = Fn::mul(Fn::from_plain(s), rinv);
      u32 p1[LN], p2[LN];
      Fn::to_plain(p1, s1);
      Fn::to_plain(p2, s2);
      ELL_UNROLL
      for (int l = 0; l < LN; l++) { p1[l] = ok ? p1[l] : 0u; p2[l] = ok ? p2[l] : 0u; }
      store_be<LN>(s1b + i * NBYTES, p1, NBYTES);
      store_be<LN>(s2b + i * NBYTES, p2, NBYTES);
    }
  }
  // Pass 4: fold the decompression result and the infinity flag into the status; no point is
  // reported for items whose status is not RECOVER_POINT
  ELL_HD static void recover_finish(size_t i, const u8* dec_ok, const u8* inf, u8* out_xy, u8* status) {
    int st = status[i];
    if (st == RECOVER_POINT && !dec_ok[i]) st = RECOVER_THROWS;          // 'invalid point'
    else if (st == RECOVER_POINT && inf[i]) st = RECOVER_INF;
    status[i] = (u8)st;
    if (st != RECOVER_POINT) {
      ELL_NOUNROLL
      for (int b = 0; b < 2 * BYTES; b++) out_xy[i * 2 * BYTES + b] = 0;
    }
  }

  // ---- EC#sign's own nonces (ec/index.js:136-158): HmacDRBG over the curve's hash
  // (curves.js `hash:`), entropy = the private key, nonce = the truncated message, both
  // n.byteLength() bytes; drbg.generate(n.byteLength()) is repeated while the candidate,
  // truncated like a digest (_truncateToN(k, true)), is <= 1 or >= n - 1.  Writes the accepted
  // candidate bytes (what the supplied-nonce pipeline below takes), zeros if none was accepted
  // within 16 draws (never in practice; the item is then reported as not signed).
  typedef typename std::conditional<CV::ID == CURVE_P384, Sha384,
          typename std::conditional<CV::ID == CURVE_P521, Sha512, Sha256>::type>::type SignHash;
  // the private key as the reference's DRBG sees it: KeyPair#_importPrivate reduces it mod n
  // (ec/key.js:91-96) before EC#sign takes getPrivate().toArray('be', n.byteLength())
  ELL_HD static void load_priv_mod_n(u32 (&d)[LN], const u8* priv, size_t i) {
    u32 t[LN];
    load_be<LN>(t, priv + i * NBYTES, NBYTES);
    Fn::to_plain(d, Fn::from_plain(t));
  }
  ELL_HD static void det_nonce(size_t i, const u8* hash, int hash_len, int shift, const u8* priv,
                               u8* nonce_out) {
    u32 e[LN], nn[LN], nm1[LN], one1[LN];
    load_hash(e, hash + i * (size_t)hash_len, hash_len, shift);
    ELL_UNROLL
    for (int l = 0; l < LN; l++) { nn[l] = C::n[l]; one1[l] = l == 0 ? 1u : 0u; }
    bn_sub<LN>(nm1, nn, one1);
    {
      u32 t[LN];
      u32 br = bn_sub<LN>(t, e, nn);               // msg >= n -> msg - n (:106-107)
      bn_select<LN>(e, br == 0, t, e);
    }
    if constexpr (std::is_same<SignHash, Sha256>::value && NBYTES % 4 == 0 && NBYTES <= 32) {
      // word-oriented generator: n.byteLength() is a whole number of words, one V per draw
      constexpr int NW = NBYTES / 4;
      u32 d[LN], seed[2 * NW];
      load_priv_mod_n(d, priv, i);
      ELL_UNROLL
      for (int w = 0; w < NW; w++) { seed[w] = d[NW - 1 - w]; seed[NW + w] = e[NW - 1 - w]; }
      HmacDrbg256<2 * NW> g;
      g.init(seed);
      u32 v[8], k[LN];
      bool done = false;
      ELL_NOUNROLL
      for (int it = 0; it < 16 && !done; it++) {
        if (it) g.reseed();
        g.draw(v);
        ELL_UNROLL
        for (int w = 0; w < LN; w++) k[w] = w < NW ? v[NW - 1 - w] : 0u;      // no shift: 8 NBYTES == bit length of n
        done = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1);
      }
      ELL_UNROLL
      for (int w = 0; w < LN; w++) k[w] = done ? k[w] : 0u;
      store_be<LN>(nonce_out + i * NBYTES, k, NBYTES);
    } else if constexpr (std::is_same<SignHash, Sha384>::value && NBYTES == 48) {
      // p384: the same with 64-bit words (SHA-384: 6-word K / V, one V per 48-byte draw)
      u32 d[LN];
      load_priv_mod_n(d, priv, i);
      u64 seed[12];
      ELL_UNROLL
      for (int w = 0; w < 6; w++) {
        seed[w] = ((u64)d[11 - 2 * w] << 32) | d[10 - 2 * w];
        seed[6 + w] = ((u64)e[11 - 2 * w] << 32) | e[10 - 2 * w];
      }
      HmacDrbg512<6, 12> g;
      g.init(seed);
      u64 v[6];
      u32 k[LN];
      bool done = false;
      ELL_NOUNROLL
      for (int it = 0; it < 16 && !done; it++) {
        if (it) g.reseed();
        g.draw(v);
        ELL_UNROLL
        for (int w = 0; w < 6; w++) { k[11 - 2 * w] = (u32)(v[w] >> 32); k[10 - 2 * w] = (u32)v[w]; }
        done = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1);
      }
      ELL_UNROLL
      for (int w = 0; w < LN; w++) k[w] = done ? k[w] : 0u;
      store_be<LN>(nonce_out + i * NBYTES, k, NBYTES);
    } else if constexpr (std::is_same<SignHash, Sha512>::value && NBYTES > 64 && NBYTES <= 128) {
      // p521: 66-byte entropy, nonce and draws -- word-oriented state, byte-granular tails
      u8 sb[1 + 2 * NBYTES], kb[NBYTES];
      {
        u32 d[LN];
        load_priv_mod_n(d, priv, i);
        store_be<LN>(sb + 1, d, NBYTES);
      }
      store_be<LN>(sb + 1 + NBYTES, e, NBYTES);
      HmacDrbg512Bytes g;
      g.init(sb, 2 * NBYTES);
      bool done = false;
      ELL_NOUNROLL
      for (int it = 0; it < 16 && !done; it++) {
        u64 v[16];
        if (it) g.reseed(sb);
        g.draw2(v);
        ELL_UNROLL
        for (int b = 0; b < NBYTES; b++) kb[b] = (u8)(v[b >> 3] >> (56 - 8 * (b & 7)));
        u32 k[LN];
        load_nonce(k, kb);
        done = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1);
      }
      ELL_NOUNROLL
      for (int b = 0; b < NBYTES; b++) nonce_out[i * NBYTES + b] = done ? kb[b] : (u8)0;
    } else {
      u8 eb[NBYTES], kb[NBYTES], pb[NBYTES];
      store_be<LN>(eb, e, NBYTES);
      {
        u32 d[LN];
        load_priv_mod_n(d, priv, i);
        store_be<LN>(pb, d, NBYTES);
      }
      HmacDrbg<SignHash> g;
      g.init(pb, NBYTES, eb, NBYTES);
      bool done = false;
      ELL_NOUNROLL
      for (int it = 0; it < 16 && !done; it++) {
        if (it) g.reseed();
        g.draw(kb, NBYTES);
        u32 k[LN];
        load_nonce(k, kb);
        done = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1);
      }
      ELL_NOUNROLL
      for (int b = 0; b < NBYTES; b++) nonce_out[i * NBYTES + b] = done ? kb[b] : (u8)0;
    }
  }

  // ---- ECDSA sign for supplied nonces (ec/index.js:110-186, one pass of its loop) ----
  // nonce K (NBYTES big-endian, as HmacDRBG#generate returns it) -> k = _truncateToN(K, true):
  // K reaches it as a BN, so the shift is 8*byteLength(value) - n.bitLength() when positive
  // (:153-156) -- only p521 (521-bit n in 66 bytes) ever shifts.
  ELL_HD static void load_nonce(u32 (&k)[LN], const u8* kb) {
    u32 t[LN];
    load_be<LN>(t, kb, NBYTES);
    constexpr int SH = 8 * NBYTES - C::NBITS;             // 0 except for p521 (7)
    if (SH > 0) {
      // top byte non-zero <=> byteLength == NBYTES <=> shift
      bool full = (t[(NBYTES - 1) / 4] >> (8 * ((NBYTES - 1) % 4))) != 0;
      ELL_UNROLL
      for (int i = 0; i < LN; i++) {
        u32 hi = i + 1 < LN ? t[i + 1] : 0u;
        u32 sh = SH > 0 ? ((t[i] >> (SH & 31)) | (hi << ((32 - SH) & 31))) : t[i];
        k[i] = full ? sh : t[i];
      }
    } else {
      bn_copy<LN>(k, t);
    }
  }
  // pass A: k*G (comb) for every nonce -> Jacobian scratch (then normalize -> affine)
  ELL_HD static void sign_mul(size_t i, size_t n, const u8* nonces, const A* comb, u32* jac) {
    u32 k[LN], kk[L];
    load_nonce(k, nonces + i * NBYTES);
    ELL_UNROLL
    for (int l = 0; l < L; l++) kk[l] = l < LN ? k[l] : 0u;
    J r = LD::template comb_mul<L, COMB_W, COMB_BITS, COMB_SIGNED>(kk, comb);
    store_jac(jac, n, i, r);
  }
  // k^-1 R mod n for item i (1 where the nonce is outside [2, n - 2]: sign_finish rejects it),
  // limb-major -- the parted sign's inversion, on a wave beside the one that computes k*G
  ELL_HD static void sign_kinv(size_t i, size_t n, const u8* nonces, u32* kinv, bool writer) {
    u32 nn[LN], nm1[LN], one1[LN], k[LN];
    ELL_UNROLL
    for (int l = 0; l < LN; l++) { nn[l] = C::n[l]; one1[l] = l == 0 ? 1u : 0u; }
    bn_sub<LN>(nm1, nn, one1);
    load_nonce(k, nonces + i * NBYTES);
    const bool ok = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1);
    Nl v = Fn::inv(fe_select<Fn>(ok, Fn::from_plain(k), Fn::one()));
    if (writer) {
      ELL_UNROLL
      for (int l = 0; l < LN; l++) kinv[(size_t)l * n + i] = v.v[l];
    }
  }
  // pass B: thread t finishes items t, t+T, ...: r = x mod n, s = k^-1 (z + r d) mod n with
  // one inversion per K items, recovery parameter, optional low-s form.  ok = 0 where the
  // reference would go on to its next nonce (k <= 1, k >= n-1, k*G = O, r = 0, s = 0).
  ELL_HD static void sign_finish(size_t t, size_t T, size_t n, int K, const u8* hash, int hash_len,
                                 int shift, const u8* priv, const u8* nonces, const u8* kg_xy,
                                 const u8* kg_inf, int canonical, u32* pre, u8* out_r, u8* out_s,
                                 u8* out_recid, u8* out_ok, const u32* kinv_in = nullptr) {
    // kinv_in (small batches, K = 1): k^-1 R mod n per item (limb-major), computed BESIDE k*G by a
    // wave of its own (sign_kinv below) -- the inversion is then off this kernel's chain
    u32 nn[LN], nm1[LN], one1[LN];
    ELL_UNROLL
    for (int l = 0; l < LN; l++) { nn[l] = C::n[l]; one1[l] = l == 0 ? 1u : 0u; }
    bn_sub<LN>(nm1, nn, one1);
    Nl acc = Fn::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      u32 k[LN];
      load_nonce(k, nonces + i * NBYTES);
      bool ok = !bn_is_zero<LN>(k) && !bn_eq<LN>(k, one1) && !bn_geq<LN>(k, nm1) && kg_inf[i] == 0;
      out_ok[i] = ok ? 1 : 0;
      Nl km = fe_select<Fn>(ok, Fn::from_plain(k), Fn::one());
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = Fn::mul(acc, km);
    }
    Nl inv = Fn::one();
    if (kinv_in) {
      ELL_UNROLL
      for (int l = 0; l < LN; l++) inv.v[l] = kinv_in[(size_t)l * n + t];      // (K = 1: item t, prefix = one)
    } else {
      inv = Fn::inv(acc);
    }
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      u32 k[LN], e[LN], d[LN], x[L], y[L];
      load_nonce(k, nonces + i * NBYTES);
      load_hash(e, hash + i * (size_t)hash_len, hash_len, shift);
      load_be<LN>(d, priv + i * NBYTES, NBYTES);
      load_be<L>(x, kg_xy + i * 2 * BYTES, BYTES);
      load_be<L>(y, kg_xy + i * 2 * BYTES + BYTES, BYTES);
      bool ok = out_ok[i] != 0;
      Nl km = fe_select<Fn>(ok, Fn::from_plain(k), Fn::one());
      Nl pr;
      ELL_UNROLL
      for (int l = 0; l < LN; l++) pr.v[l] = pre[(size_t)l * n + i];
      Nl kinv = Fn::mul(inv, pr);
      inv = Fn::mul(inv, km);
      // r = x mod n: x < p < 2n for every preset, so at most one subtraction
      static_assert(LN <= L, "order wider than field");
      u32 nx[L], xr[L];
      ELL_UNROLL
      for (int l = 0; l < L; l++) nx[l] = l < LN ? C::n[l] : 0u;
      u32 br = bn_sub<L>(xr, x, nx);
      bool wrapped = br == 0;                         // x >= n
      u32 r[LN];
      ELL_UNROLL
      for (int l = 0; l < LN; l++) r[l] = wrapped ? xr[l] : x[l];
      // s = k^-1 (e + r d) with three Montgomery products instead of six: (r) * (d R) / R = r d as a
      // PLAIN residue, e joins it plain (e < 2^bits(n) < 2n: one conditional subtraction), and
      // (e + r d) * (k^-1 R) / R is s, plain -- r, e are never converted in, s never out
      Nl rp, ep;
      bn_copy<LN>(rp.v, r);
      {
        u32 en[LN];
        u32 be = bn_sub<LN>(en, e, nn);
        ELL_UNROLL
        for (int l = 0; l < LN; l++) ep.v[l] = be ? e[l] : en[l];
      }
      Nl sm = Fn::mul(Fn::add(Fn::mul(rp, Fn::from_plain(d)), ep), kinv);
      u32 sp[LN];
      bn_copy<LN>(sp, sm.v);
      ok = ok && !bn_is_zero<LN>(r) && !bn_is_zero<LN>(sp);
      u32 recid = (y[0] & 1u) | (wrapped ? 2u : 0u);
      // low-s form: s > n >> 1  ->  s = n - s, recid ^= 1
      u32 nh[LN], ns[LN];
      ELL_UNROLL
      for (int l = 0; l < LN; l++) nh[l] = (nn[l] >> 1) | (l + 1 < LN ? nn[l + 1] << 31 : 0u);
      bool high = canonical && !bn_geq<LN>(nh, sp);
      bn_sub<LN>(ns, nn, sp);
      ELL_UNROLL
      for (int l = 0; l < LN; l++) sp[l] = high ? ns[l] : sp[l];
      if (high) recid ^= 1u;
      if (!ok) { bn_zero<LN>(r); bn_zero<LN>(sp); recid = 0; }
      store_be<LN>(out_r + i * NBYTES, r, NBYTES);
      store_be<LN>(out_s + i * NBYTES, sp, NBYTES);
      out_recid[i] = (u8)recid;
      out_ok[i] = ok ? 1 : 0;
    }
  }

  // JPoint#eqXToP (short.js:908-925): X == r*Z^2, retry with r+n while < p.
  // (for every preset p < 2n, so one retry at most)
  ELL_HD static bool eq_x_to_p(const J& p, const u32 (&r)[LN]) {
    static_assert(LN <= L, "order wider than field");
    u32 rx[L], pp[L];
    ELL_UNROLL
    for (int i = 0; i < L; i++) { rx[i] = i < LN ? r[i] : 0u; pp[i] = C::p[i]; }
    El zz = F::sqr(p.Z);
    if (F::eq(p.X, F::mul(F::from_plain(rx), zz))) return true;
    u32 nn[L], rn[L];
    ELL_UNROLL
    for (int i = 0; i < L; i++) nn[i] = i < LN ? C::n[i] : 0u;
    u32 cy = bn_add<L>(rn, rx, nn);
    if (cy || bn_geq<L>(rn, pp)) return false;
    return F::eq(p.X, F::mul(F::from_plain(rn), zz));
  }

  // ---- the small-grid form of pass 2, in two kernels (secp256k1) ---------------------------
  // A batch that leaves two or three waves per SIMD resident (one GPU's share of BASELINE's 2^20
  // over eight) is latency-bound: every lane is ONE dependent chain, s^-1 -> u2 -> digits ->
  // ladder.  But the window table of Q -- a twentieth of the chain -- needs only the key, not
  // u2: ecdsa_table builds it BESIDE ecdsa_prep -- other workgroups of the same launch
  // (FnEcdsaPrepTable, Engine::ecdsa_chunk) -- and ecdsa_ladder starts from the finished table.
  // The table's common Z (zg) travels in the table's last scratch slot.
  template <bool WIDE>
  ELL_HD static void ecdsa_table(size_t i, size_t n, const u8* pub_xy, VT* tbl_all) {
    static_assert(ENDO, "the split verify is the endomorphism curve's");
    typedef Endo<WIDE> E;
    A q = load_affine(pub_xy, i);
    VT* tbl = tbl_all + i * stride<WIDE>();
    El zg;
    LD::template build_table_odd8<E::NE>(tbl, q, zg);
    tbl[2 * E::NE - 1].x = zg;
  }
  template <bool WIDE>
  ELL_HD static void ecdsa_ladder(size_t i, size_t n, const u32* u12, const u8* valid, const u8* rs,
                                  const u8* pub_xy, const A* comb, const VT* tbl_all,
                                  const DigitStore& ds, u8* out_ok, u8* out_st) {
    typedef Endo<WIDE> E;
    u32 u1[L], u2[L], r[LN];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u2[l] = l < LN ? u12[(size_t)(1 * LN + l) * n + i] : 0u;
    const VT* tbl = tbl_all + i * stride<WIDE>();
    u32 k1[5], k2[5];
    bool n1, n2;
    glv_split<true>(u2, k1, n1, k2, n2);
    recode_odd_w4<5, E::NW, E::WB>(k1, ds, 0, 2);
    recode_odd_w4<5, E::NW, E::WB>(k2, ds, 1, 2);
    const u32 negmask = (n1 ? 1u : 0u) | (n2 ? 2u : 0u);
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<2, E::NW, true, WIDE, E::WB>(ds, tbl, negmask, 0u, inf, &beta);
    b.Z = F::mul(b.Z, tbl[2 * E::NE - 1].x);             // back from the table's isomorphic curve
    ELL_UNROLL
    for (int l = 0; l < L; l++) u1[l] = l < LN ? u12[(size_t)(0 * LN + l) * n + i] : 0u;
    J p = LD::template comb_add<L, COMB_W, COMB_BITS, WIDE, COMB_SIGNED>(b, inf, u1, comb);
    load_be<LN>(r, rs + i * NBYTES, NBYTES);
    bool ok = !G::is_inf(p);
    ok = ok && eq_x_to_p(p, r);
    const bool on = on_curve(load_affine(pub_xy, i));      // see ecdsa_main
    store_verdict(i, valid[i], on, ok, out_ok, out_st);
  }

  // ---- the parted form of pass 2 (secp256k1): one verify on THREE lanes ---------------------
  // A batch that does not give every SIMD of the device a wave -- a single EC#verify, the handful
  // a server coalesces in one tick -- is one dependent chain per item and nothing else: the
  // machine idles beside it.  R = u1*G + k1*Q + k2*(lambda Q) is a sum of three independent
  // scalar multiplications, so the parted form runs them in three DIFFERENT waves (workgroups
  // of the same launch, FnEcdsaParts): part 0 the ladder of k1 over Q's window table, part 1
  // that of k2 over the same table with beta at every lookup, part 2 the comb of u1 -- and
  // ecdsa_join adds the three Jacobian results and compares.  The chain per item falls from
  // 125 doublings + 51 additions + 12 comb additions to 125 doublings + 25 additions + two full
  // Jacobian additions; the total work rises by a tenth (which an idle machine does not feel).
  // Parts 0 and 1 leave their sums on the table's isomorphic curve (Z not yet scaled by zg).
  template <bool WIDE>
  ELL_HD static void ecdsa_half(size_t i, size_t n, int half, const u32* u12, const VT* tbl_all,
                                const DigitStore& ds, u32* jac) {
    static_assert(ENDO, "the parted verify is the endomorphism curve's");
    typedef Endo<WIDE> E;
    u32 u2[L];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u2[l] = l < LN ? u12[(size_t)(1 * LN + l) * n + i] : 0u;
    u32 k1[5], k2[5];
    bool n1, n2;
    glv_split<true>(u2, k1, n1, k2, n2);
    const bool lam = half != 0;
    ELL_UNROLL
    for (int l = 0; l < 5; l++) k1[l] = lam ? k2[l] : k1[l];
    recode_odd_w4<5, E::NW, E::WB>(k1, ds, 0, 1);
    const u32 negmask = (lam ? n2 : n1) ? 1u : 0u;
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<1, E::NW, true, WIDE, E::WB>(ds, tbl_all + i * stride<WIDE>(), negmask, 0u,
                                                               inf, &beta, lam);
    store_jac(jac, n, i, b);
  }
  ELL_HD static void ecdsa_fixed(size_t i, size_t n, const u32* u12, const A* comb, u32* jac) {
    u32 u1[L];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u1[l] = l < LN ? u12[(size_t)(0 * LN + l) * n + i] : 0u;
    J c = LD::template comb_mul<L, COMB_W, COMB_BITS, COMB_SIGNED>(u1, comb);
    store_jac(jac, n, i, c);
  }
  // jac: the three parts' results, 3 * NS * n words each
  template <bool WIDE>
  ELL_HD static void ecdsa_join(size_t i, size_t n, const u8* valid, const u8* rs, const u8* pub_xy,
                                const VT* tbl_all, const u32* jac, u8* out_ok, u8* out_st) {
    typedef Endo<WIDE> E;
    const size_t part = (size_t)3 * NS * n;
    J b = G::add(load_jac(jac, n, i), load_jac(jac + part, n, i));     // any of O, P = Q, P = -Q included
    b.Z = F::mul(b.Z, tbl_all[i * stride<WIDE>() + 2 * E::NE - 1].x);   // back from the table's isomorphic curve
    J p = G::add(b, load_jac(jac + 2 * part, n, i));
    u32 r[LN];
    load_be<LN>(r, rs + i * NBYTES, NBYTES);
    bool ok = !G::is_inf(p);
    ok = ok && eq_x_to_p(p, r);
    const bool on = on_curve(load_affine(pub_xy, i));      // see ecdsa_main
    store_verdict(i, valid[i], on, ok, out_ok, out_st);
  }

  // The join of the curves without an endomorphism (two parts: u2*Q and u1*G, computed by the
  // row layer's waves, coop_work.h CoopNist): their sum, the x test, the verdict.
  ELL_HD static void ecdsa_join2(size_t i, size_t n, const u8* valid, const u8* rs, const u8* pub_xy,
                                 const u32* jac, u8* out_ok, u8* out_st) {
    const size_t part = (size_t)3 * NS * n;
    J p = G::add(load_jac(jac, n, i), load_jac(jac + part, n, i));     // any of O, P = Q, P = -Q included
    u32 r[LN];
    load_be<LN>(r, rs + i * NBYTES, NBYTES);
    bool ok = !G::is_inf(p);
    ok = ok && eq_x_to_p(p, r);
    const bool on = on_curve(load_affine(pub_xy, i));
    store_verdict(i, valid[i], on, ok, out_ok, out_st);
  }

  // ... and Point#mul the same way: k*P = k1*P + k2*(lambda P) on two lanes.  Each half builds
  // its own copy of P's window table (no launch in front of the ladder; an idle machine does not
  // feel the second build) and returns its sum in true Jacobian coordinates; mul_join adds.
  template <bool WIDE>
  ELL_HD static void mul_half(size_t i, size_t n, int half, const u8* ks, const u8* xy, VT* tbl_all,
                              const DigitStore& ds, u32* jac) {
    static_assert(ENDO, "the parted ladder is the endomorphism curve's");
    typedef Endo<WIDE> E;
    u32 k[L];
    load_be<L>(k, ks + i * BYTES, BYTES);
    u32 k1[5], k2[5];
    bool n1, n2;
    glv_split<true>(k, k1, n1, k2, n2);
    const bool lam = half != 0;
    ELL_UNROLL
    for (int l = 0; l < 5; l++) k1[l] = lam ? k2[l] : k1[l];
    recode_odd_w4<5, E::NW, E::WB>(k1, ds, 0, 1);
    const u32 negmask = (lam ? n2 : n1) ? 1u : 0u;
    VT* tbl = tbl_all + i * stride<WIDE>();
    El zg;
    LD::template build_table_odd8<E::NE>(tbl, load_affine(xy, i), zg);
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<1, E::NW, true, WIDE, E::WB>(ds, tbl, negmask, 0u, inf, &beta, lam);
    b.Z = F::mul(b.Z, zg);
    store_jac(jac, n, i, b);
  }
  // (third part of k1*G + k2*P in the parted form: the comb of k1)
  ELL_HD static void mul_fixed_part(size_t i, size_t n, const u8* ks, const A* comb, u32* jac) {
    u32 k[L];
    load_be<L>(k, ks + i * BYTES, BYTES);
    store_jac(jac, n, i, LD::template comb_mul<L, COMB_W, COMB_BITS, COMB_SIGNED>(k, comb));
  }
  // jac: the two halves' results (3 * NS * n words each).  The join is also the call's last
  // kernel: the sum goes to affine coordinates with an inversion of its own (no Montgomery trick
  // across items: on an idle machine the chain counts, not the work) and the operand is tested
  // against the curve equation here (normalize + domain_mark of the one-lane form).
  // `with_g`: a third result, k1*G of k1*G + k2*P, follows the halves'.
  ELL_HD static void mul_join(size_t i, size_t n, const u32* jac, bool with_g, const u8* xy, u8* out_xy,
                              u8* out_inf, A* raw_aff) {
    const size_t part = (size_t)3 * NS * n;
    J p = G::add(load_jac(jac, n, i), load_jac(jac + part, n, i));
    if (with_g) p = G::add(p, load_jac(jac + 2 * part, n, i));
    const bool inf = F::is_zero(p.Z);
    El zinv = F::inv(fe_select<F>(inf, F::one(), p.Z));
    El zi2 = F::sqr(zinv);
    El x = F::mul(p.X, zi2);
    El y = F::mul(p.Y, F::mul(zi2, zinv));
    if (inf) { x = F::zero(); y = F::zero(); }
    if (raw_aff) { raw_aff[i].x = x; raw_aff[i].y = y; }
    bool on = true;
    if (out_inf) on = on_curve(load_affine(xy, i));            // (the comb build has no out_inf: its points are G)
    if (!on) { x = F::zero(); y = F::zero(); }
    if (out_xy) {
      store_fe(out_xy + i * 2 * BYTES, x);
      store_fe(out_xy + i * 2 * BYTES + BYTES, y);
    }
    if (out_inf) out_inf[i] = !on ? (u8)DOMAIN_OFF_CURVE : (inf ? (u8)1 : (u8)0);
  }

  // Pass 2: R = u1*G + u2*Q, accept iff R != O and R.x == r (mod n)
  template <bool WIDE = false>
  ELL_HD static void ecdsa_main(size_t i, size_t n, const u32* u12, const u8* valid,
                                const u8* rs, const u8* pub_xy, const A* comb, VT* tbl_all,
                                const DigitStore& ds, u8* out_ok, u8* out_st) {
    u32 u1[L], u2[L], r[LN];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u2[l] = l < LN ? u12[(size_t)(1 * LN + l) * n + i] : 0u;
    A q = load_affine(pub_xy, i);
    // u2 * Q first; u1 and r are loaded where they are used, behind compiler barriers, so that
    // they do not occupy registers across the ladder (ELL_LATE_LOADS: the 128-register build)
    bool inf;
    J b = var_ladder<WIDE>(u2, q, tbl_all + i * stride<WIDE>(), ds, inf);
#if ELL_LATE_LOADS && defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!WIDE) asm volatile("" ::: "memory");
#endif
    ELL_UNROLL
    for (int l = 0; l < L; l++) u1[l] = l < LN ? u12[(size_t)(0 * LN + l) * n + i] : 0u;
    J p = LD::template comb_add<L, COMB_W, COMB_BITS, WIDE, COMB_SIGNED>(b, inf, u1, comb);
#if ELL_LATE_LOADS && defined(__HIP_DEVICE_COMPILE__)
    if constexpr (!WIDE) asm volatile("" ::: "memory");
#endif
    load_be<LN>(r, rs + i * NBYTES, NBYTES);
    bool ok = !G::is_inf(p);
    ok = ok && eq_x_to_p(p, r);
    // The key is re-read and tested against the curve equation HERE, after the ladder, where
    // nothing but the verdict is live (a flag carried across the ladder would cost the
    // 128-register build a spill): r or s out of range -> 0 as in the reference, which returns
    // false before it touches the key (ec/index.js:199-202); a key that is not on the curve ->
    // verdict 0 with status 2, outside the engine's domain (see on_curve above); else the verdict.
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" ::: "memory");
#endif
    const bool on = on_curve(load_affine(pub_xy, i));
    store_verdict(i, valid[i], on, ok, out_ok, out_st);
  }
};

}  // namespace ell

#include "coop_work.h"
