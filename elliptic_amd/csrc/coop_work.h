// ellgpu -- per-item work of the lanes-per-item layer (coop.h): the parts of one secp256k1
// EC#verify / Point#mul, each on a WAVE of its own (one item per wave, its field elements spread
// over a 16-lane row).  Same decomposition as the one-lane parted form (work.h ecdsa_half /
// ecdsa_fixed / mul_half / mul_fixed_part): R = u1*G + k1*Q + k2*(lambda Q) as three independent
// sums in three waves, k*P = k1*P + k2*(lambda P) in two; same GLV split, same odd-digit
// recoding, same tables, same ladder templates (ladder.h run_odd_w4 / build_table_odd8 over
// Ladder<CvSecp256k1C>) -- only the field under them is the row's (coop.h FpK256C).  The results go
// to the one-lane join kernels (ecdsa_join / mul_join) in THEIR format: canonical FpK256 words.
//
// Included at the end of work.h (it borrows Work<CvSecp256k1>'s geometry and I/O layouts).
#pragma once

#include "coop.h"
#include "coop_mont.h"
#include "coop_wide.h"

namespace ell {

// k*G over the one-lane kernels' comb table (ladder.h comb_add with the entries read through
// CW::load_entry): CW = the row layer's work struct (CoopK256 / CoopNist<...>)
template <class CW, int LK>
ELL_HD typename CW::J coop_comb_mul(const u32 (&k)[LK], const typename CW::W1::A* comb) {
  typedef typename CW::W1 W1;
  typedef typename CW::A A;
  typedef typename CW::G G;
  typedef typename CW::LD LD;
  constexpr bool SIGNED = W1::COMB_SIGNED;
  const int CB = SIGNED ? comb_bits_of(comb) : W1::COMB_BITS;        // (a signed table carries its window width)
  const int W = SIGNED ? comb_windows(32 * LK, CB) : W1::COMB_W;
  const u32 MASK = (1u << CB) - 1u;
  const u32 HALF = 1u << (CB - 1);
  const u32 PER = SIGNED ? HALF : MASK;
  u32 kk[LK];
  bn_copy<LK>(kk, k);
  u32 carry = 0;
  typename CW::J acc = G::infinity();
  bool inf = true;
  ELL_NOUNROLL
  for (int w = 0; w < W; w++) {
    u32 d = (kk[0] & MASK) + carry;
    ELL_UNROLL
    for (int j = 0; j < LK - 1; j++) kk[j] = (kk[j] >> CB) | (kk[j + 1] << (32 - CB));
    kk[LK - 1] >>= CB;
    bool neg = false;
    if (SIGNED) {
      neg = d > HALF;
      carry = neg ? 1u : 0u;
      d = neg ? (MASK + 1u) - d : d;
    }
    if (d != 0) {                                        // wave-uniform: one item per wave
      const typename W1::A* e = comb + ((size_t)w * PER + (d - 1u));
      auto fetch = [&]() -> A {
        A q = CW::load_entry(e);
        if (SIGNED) q.y = LD::cneg_y(q.y, neg);
        return q;
      };
      acc = G::add_mixed_lean(acc, fetch(), inf, fetch);
    }
  }
  return acc;
}

// EC#sign's k*G for a handful of items (work.h sign_mul + normalize in one unit): the comb on the
// row layer, then THIS item's own inversion -- no Montgomery trick across items, on an idle machine
// the chain counts -- and the affine point as the one-lane sign_finish reads it (big-endian x || y,
// infinity flag).  CW = CoopK256 / CoopNist<...>.
// NONCE: the scalar is an HmacDRBG draw and goes through _truncateToN(K, true) as EC#sign's does
// (W1::load_nonce: p521 shifts a full-width draw by 7 bits); else it is Point#mul's scalar, any
// BYTES-byte value as it stands.
template <class CW, bool NONCE = true>
ELL_HD void coop_sign_point(size_t i, const u8* nonces, const typename CW::W1::A* comb, u8* kg_xy, u8* kg_inf) {
  typedef typename CW::W1 W1;
  typedef typename CW::F F;
  constexpr int L = W1::L, LN = W1::LN, BYTES = W1::BYTES;
  u32 kk[L];
  if constexpr (NONCE) {
    u32 k[LN];
    W1::load_nonce(k, nonces + i * W1::NBYTES);
    ELL_UNROLL
    for (int l = 0; l < L; l++) kk[l] = l < LN ? k[l] : 0u;
  } else {
    load_be<L>(kk, nonces + i * BYTES, BYTES);
  }
  typename CW::J r = coop_comb_mul<CW>(kk, comb);
  const bool inf = F::is_zero(r.Z);
  typename F::El zi = F::inv(r.Z);                      // inv(0) = 0
  typename F::El zi2 = F::sqr(zi);
  u32 x[L], y[L];
  F::to_plain(x, F::mul(r.X, zi2));
  F::to_plain(y, F::mul(r.Y, F::mul(zi2, zi)));
  if (CW::writer()) {
    u8* o = kg_xy + i * 2 * BYTES;
    if (inf) { ELL_UNROLL for (int l = 0; l < L; l++) { x[l] = 0; y[l] = 0; } }
    store_be<L>(o, x, BYTES);
    store_be<L>(o + BYTES, y, BYTES);
    kg_inf[i] = inf ? 1 : 0;
  }
}

// PR = false: one item per WAVE (FpK256C); PR = true: one item per ROW, four items per wave
// (FpK256R -- coop.h; batches between Tuning::coop_grid and Tuning::row_grid)
template <bool PR>
struct CoopK256T {
  typedef CvSecp256k1CT<PR> CV;
  typedef FpK256CT<PR> F;
  typedef typename F::El El;
  typedef ShortOps<CV> G;
  typedef Ladder<CV> LD;
  typedef Jac<F> J;
  typedef Aff<F> A;
  typedef Work<CvSecp256k1> W1;                      // the one-lane layer: geometry, table and result formats
  typedef typename W1::template Endo<true> E;                          // the small-grid tuning's windows (5 bits)
  // the row layer reads and writes the SATURATED field's tables and results; the ELL_K256_LAZY
  // build (one-lane kernels on fpk256l.h) keeps its small batches on the one-lane parts
  static constexpr bool AVAILABLE = std::is_same<CvSecp256k1::F, FpK256>::value;
  // table slots of one lane: the odd multiples + build_table_odd8's ratios
  static constexpr int SLOTS = 2 * E::NE;
  // bytes of row memory a unit needs (k_run_coop's LDS / the host loop's stack): SLOTS entries of
  // every lane of the row
  // (one item per row: the device keeps a table per LANE of the wave -- 64 of them; host passes walk the
  // rows one after the other through the same memory)
  static constexpr int TLANES = PR ? 64 : 16;
  static constexpr int ROW_BYTES = SLOTS * TLANES * 16;
  static_assert(sizeof(A) * SLOTS * (F::CL == 1 ? TLANES : 1) <= (size_t)ROW_BYTES, "row memory too small");

  // this lane's table: SLOTS entries of its own
  ELL_HD static A* lane_table(void* row_mem) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (A*)row_mem + (size_t)(threadIdx.x & (unsigned)(TLANES - 1)) * SLOTS;
#else
    return (A*)row_mem;
#endif
  }
  // does this thread store the results (wave-uniform; per row when every row has an item of its own)?
  ELL_HD static bool writer() {
#if defined(__HIP_DEVICE_COMPILE__)
    return PR ? (threadIdx.x & 15u) == 0 : threadIdx.x == 0;
#else
    return true;
#endif
  }

  // a point in the one-lane kernels' memory format (Aff<FpK256>: x, y as eight plain words) -> the row
  ELL_HD static A load_entry(const typename W1::A* e) {
    A q;
    q.x = F::load_words(e->x.v);
    q.y = F::load_words(e->y.v);
    return q;
  }
  // big-endian coordinates at the C ABI -> the row
  ELL_HD static A load_affine(const u8* xy, size_t i) {
    u32 tx[8], ty[8];
    load_be<8>(tx, xy + i * 64, 32);
    load_be<8>(ty, xy + i * 64 + 32, 32);
    A q;
    q.x = F::from_plain(tx);
    q.y = F::from_plain(ty);
    return q;
  }
  // Jacobian result -> the join kernels' format (W1::store_jac: canonical FpK256 words, limb-major)
  ELL_HD static void store_jac(u32* jac, size_t n, size_t i, const J& p) {
    u32 x[8], y[8], z[8];
    F::to_plain(x, p.X);
    F::to_plain(y, p.Y);
    F::to_plain(z, p.Z);
    if (writer()) {
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        jac[(size_t)(0 * 8 + l) * n + i] = x[l];
        jac[(size_t)(1 * 8 + l) * n + i] = y[l];
        jac[(size_t)(2 * 8 + l) * n + i] = z[l];
      }
    }
  }
  ELL_HD static El load_beta() {
    u32 b[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) b[i] = consts::SECP256K1_C::beta[i];
    return F::from_plain(b);
  }
  // one GLV half of k as odd digits in ds; returns the half's sign
  ELL_HD static bool half_digits(const u32 (&k)[8], bool lam, const DigitStore& ds) {
    u32 k1[5], k2[5];
    bool n1, n2;
    glv_split<true>(k, k1, n1, k2, n2);
    ELL_UNROLL
    for (int l = 0; l < 5; l++) k1[l] = lam ? k2[l] : k1[l];
    recode_odd_w4<5, E::NW, E::WB>(k1, ds, 0, 1);
    return lam ? n2 : n1;
  }

  // ---- EC#verify, parts 0 and 1: k1*Q or k2*(lambda Q) over the table ecdsa_prep_table built ----
  // (work.h ecdsa_half; the sum stays on the table's isomorphic curve, ecdsa_join scales Z by zg)
  ELL_HD static void ecdsa_half(size_t i, size_t n, int half, const u32* u12, const typename W1::VT* tbl_all,
                                const DigitStore& ds, u32* jac, void* row_mem) {
    u32 u2[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++) u2[l] = u12[(size_t)(1 * W1::LN + l) * n + i];
    const bool lam = half != 0;
    const u32 negmask = half_digits(u2, lam, ds) ? 1u : 0u;
    A* tbl = lane_table(row_mem);
    const typename W1::VT* src = tbl_all + i * W1::template stride<true>();
    ELL_NOUNROLL
    for (int e = 0; e < E::NE; e++) tbl[e] = load_entry(src + e);
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<1, E::NW, true, false, E::WB>(ds, tbl, negmask, 0u, inf, &beta, lam);
    store_jac(jac, n, i, b);
  }

  // ---- EC#verify, in front of the parts: the window table of Q on a wave (work.h ecdsa_table) ----
  // The same table -- build_table_odd8 is the same template, its entries and the common Z are the
  // same field values -- written in the one-lane kernels' format (plain words; zg in the table's
  // last scratch slot), where ecdsa_half / ecdsa_join read it.  Runs BESIDE the scalar-field prep
  // (FnEcdsaPrepTableC): the one-lane table build was the longer of the two (86 against 55 us).
  // (N = 8: the saturated field's words.  The ELL_K256_LAZY build's nine-limb elements keep their
  // small batches off the row layer -- AVAILABLE is false there -- but this must still compile)
  template <int N>
  ELL_HD static void store_words(u32 (&dst)[N], const El& a) {
    u32 w[8];
    F::to_plain(w, a);
    if (writer()) {
      ELL_UNROLL
      for (int l = 0; l < N; l++) dst[l] = l < 8 ? w[l] : 0u;
    }
  }
  ELL_HD static void ecdsa_table(size_t i, const u8* pub_xy, typename W1::VT* tbl_all, void* row_mem) {
    A* tbl = lane_table(row_mem);
    El zg;
    LD::template build_table_odd8<E::NE>(tbl, load_affine(pub_xy, i), zg);
    typename W1::VT* dst = tbl_all + i * W1::template stride<true>();
    ELL_NOUNROLL
    for (int e = 0; e < E::NE; e++) {
      store_words(dst[e].x.v, tbl[e].x);
      store_words(dst[e].y.v, tbl[e].y);
    }
    store_words(dst[2 * E::NE - 1].x.v, zg);
  }

  // ---- fixed base: k*G over the one-lane comb table ----
  ELL_HD static J comb_mul(const u32 (&k)[8], const typename W1::A* comb) { return coop_comb_mul<CoopK256T<PR>>(k, comb); }
  // EC#verify, part 2: u1*G
  ELL_HD static void ecdsa_fixed(size_t i, size_t n, const u32* u12, const typename W1::A* comb, u32* jac) {
    u32 u1[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++) u1[l] = u12[(size_t)(0 * W1::LN + l) * n + i];
    store_jac(jac, n, i, comb_mul(u1, comb));
  }

  // ---- Point#mul, one half: builds its own window table in row memory (work.h mul_half) ----
  ELL_HD static void mul_half(size_t i, size_t n, int half, const u8* ks, const u8* xy, const DigitStore& ds,
                              u32* jac, void* row_mem) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    const bool lam = half != 0;
    const u32 negmask = half_digits(k, lam, ds) ? 1u : 0u;
    A* tbl = lane_table(row_mem);
    El zg;
    LD::template build_table_odd8<E::NE>(tbl, load_affine(xy, i), zg);
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<1, E::NW, true, false, E::WB>(ds, tbl, negmask, 0u, inf, &beta, lam);
    b.Z = F::mul(b.Z, zg);
    store_jac(jac, n, i, b);
  }
  // third part of k1*G + k2*P: the comb of k1
  ELL_HD static void mul_fixed_part(size_t i, size_t n, const u8* ks, const typename W1::A* comb, u32* jac) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    store_jac(jac, n, i, comb_mul(k, comb));
  }

  // ---- ShortCurve#pointFromX (short.js:187-204) of one item on a wave ----
  // a^((p+1)/4) with FpK256::sqrt's addition chain (253 S + 13 M; bn.js Red#sqrt takes the same
  // power): every step is a product of the row field, 2.2-2.4 x shorter than the one-lane one
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = F::sqr(a);
    return a;
  }
  ELL_HD static El sqrt(const El& a) {
    El x2 = F::mul(F::sqr(a), a);
    El x3 = F::mul(F::sqr(x2), a);
    El x6 = F::mul(sqr_n(x3, 3), x3);
    El x9 = F::mul(sqr_n(x6, 3), x3);
    El x11 = F::mul(sqr_n(x9, 2), x2);
    El x22 = F::mul(sqr_n(x11, 11), x11);
    El x44 = F::mul(sqr_n(x22, 22), x22);
    El x88 = F::mul(sqr_n(x44, 44), x44);
    El x176 = F::mul(sqr_n(x88, 88), x88);
    El x220 = F::mul(sqr_n(x176, 44), x44);
    El x223 = F::mul(sqr_n(x220, 3), x3);
    El t = F::mul(sqr_n(x223, 23), x22);
    t = F::mul(sqr_n(t, 6), x2);
    return sqr_n(t, 2);
  }
  // (work.h lift_x / decompress: the same results -- x reduced, y with the requested parity, zeros
  // and ok = 0 where x is no abscissa of the curve); xw: the wave-uniform plain words of x
  ELL_HD static void decompress_words(size_t i, const u32 (&xw)[8], bool want_odd, u8* out_xy, u8* out_ok) {
    const El x = F::from_plain(xw);
    u32 seven[8] = {7u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const El rhs = F::add(F::mul(F::sqr(x), x), F::from_plain(seven));
    const El y = sqrt(rhs);
    const bool ok = F::eq(F::sqr(y), rhs);
    u32 xp[8], yp[8], yn[8], pp[8];
    F::to_plain(xp, x);
    F::to_plain(yp, y);
    FpK256::get_p(pp);
    bn_sub<8>(yn, pp, yp);                              // p - y (y = 0 has no negative to take: no such point on this curve)
    const bool flip = ((yp[0] & 1u) != 0) != want_odd && !bn_is_zero<8>(yp);
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {
      yp[l] = flip ? yn[l] : yp[l];
      if (!ok) { xp[l] = 0; yp[l] = 0; }
    }
    if (writer()) {
      store_be<8>(out_xy + i * 64, xp, 32);
      store_be<8>(out_xy + i * 64 + 32, yp, 32);
      out_ok[i] = ok ? 1 : 0;
    }
  }
  ELL_HD static void decompress(size_t i, const u8* xs, const u8* odd, u8* out_xy, u8* out_ok) {
    u32 xw[8];
    load_be<8>(xw, xs + i * 32, 32);
    decompress_words(i, xw, odd[i] != 0, out_xy, out_ok);
  }
  // EC#recoverPubKey's R (ec/index.js:243-250) straight from (r, j): x = r, or r + n for the second
  // candidate -- what recover_prep writes into xs / odd, so that the square root does not wait for
  // the inversion of r beside it.  (Where recover_prep's status is not RECOVER_POINT the result is
  // not looked at.)
  ELL_HD static void recover_point(size_t i, const u8* rs, const u8* recid, u8* out_xy, u8* out_ok) {
    u32 r[8], rn[8], nn[8];
    load_be<8>(r, rs + i * 32, 32);
    ELL_UNROLL
    for (int l = 0; l < 8; l++) nn[l] = consts::SECP256K1_C::n[l];
    bn_add<8>(rn, r, nn);
    const u32 jj = recid[i];
    const bool second = (jj >> 1) != 0;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) r[l] = second ? rn[l] : r[l];
    decompress_words(i, r, (jj & 1u) != 0, out_xy, out_ok);
  }
};
typedef CoopK256T<false> CoopK256;
typedef CoopK256T<true> CoopK256R;


// ---- the NIST curves up to 256 bits on the row layer (coop_mont.h) --------------------------------
// No endomorphism: a verify is TWO parts -- u2*Q by the odd-digit ladder over a table of its own
// (work.h var_ladder's plain branch: co-Z table on the isomorphic curves, one inversion to map it
// back to the curve whose a = -3 doubling the ladder needs) and u1*G by the comb -- joined by the
// one-lane ecdsa_join2; Point#mul is the ladder alone, normalised by the one-lane kernel.
// (p384 and p521 -- 14 / 19 digits: 27 / 37 product columns do not fit a 16-lane row -- take the
// WIDE field, an element over the lanes of the whole wave: coop_wide.h FpFoldW)
// (FOLD: reduction by folding with constant vectors instead of the word-serial Montgomery chain --
// coop_mont.h FpFoldC for p256 in the 16-lane row; the wide layer folds by construction)
#ifndef ELL_P256_FOLD
#define ELL_P256_FOLD 1
#endif
template <class CV1>
struct CoopConsts { static constexpr bool AVAILABLE = false, WIDE = false, FOLD = false; typedef consts::COOP_P256 MC; };
template <> struct CoopConsts<CvP192> { static constexpr bool AVAILABLE = true, WIDE = false, FOLD = false; typedef consts::COOP_P192 MC; };
template <> struct CoopConsts<CvP224> { static constexpr bool AVAILABLE = true, WIDE = false, FOLD = false; typedef consts::COOP_P224 MC; };
template <> struct CoopConsts<CvP256> { static constexpr bool AVAILABLE = true, WIDE = false, FOLD = ELL_P256_FOLD != 0; typedef consts::COOP_P256 MC; };
template <> struct CoopConsts<CvP384> { static constexpr bool AVAILABLE = true, WIDE = true, FOLD = true; typedef consts::COOPW_P384 MC; };
template <> struct CoopConsts<CvP521> { static constexpr bool AVAILABLE = true, WIDE = true, FOLD = true; typedef consts::COOPW_P521 MC; };
template <bool WIDE, bool FOLD, class MC, class F1>
struct CoopField { typedef FpMontC<MC, F1> type; };
template <class MC, class F1>
struct CoopField<false, true, MC, F1> { typedef FpFoldC<MC, F1> type; };
template <bool FOLD, class MC, class F1>
struct CoopField<true, FOLD, MC, F1> { typedef FpFoldW<MC, F1> type; };

template <class CV1>
struct CoopNist {
  typedef typename CoopConsts<CV1>::MC MC;
  static constexpr bool AVAILABLE = CoopConsts<CV1>::AVAILABLE;
  static constexpr bool WIDE = CoopConsts<CV1>::WIDE;
  static constexpr int LANES = WIDE ? 64 : 16;       // lanes an element lives on (a row, or the wave)
  struct CV {
    typedef typename CoopField<CoopConsts<CV1>::WIDE, CoopConsts<CV1>::FOLD, MC, typename CV1::F>::type F;
    typedef typename CV1::Fn Fn;
    typedef typename CV1::C C;
    static constexpr int A_KIND = CV1::A_KIND;
    static constexpr bool ENDO = false;
    static constexpr bool JTABLE = false;
    static constexpr int ID = CV1::ID;
  };
  typedef typename CV::F F;
  typedef typename F::El El;
  typedef ShortOps<CV> G;
  typedef Ladder<CV> LD;
  typedef Jac<F> J;
  typedef Aff<F> A;
  typedef Work<CV1> W1;
  static constexpr int L = W1::L;
  static constexpr int NE = W1::PLAIN_NE, NW = W1::PLAIN_NW, WB = W1::PLAIN_WB;
  static constexpr int SLOTS = 2 * NE;
  static constexpr int ROW_BYTES = SLOTS * LANES * 16;
  static_assert(sizeof(A) * SLOTS * (F::CL == 1 ? LANES : 1) <= (size_t)ROW_BYTES, "row memory too small");

  ELL_HD static A* lane_table(void* row_mem) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (A*)row_mem + (size_t)(threadIdx.x & (unsigned)(LANES - 1)) * SLOTS;
#else
    return (A*)row_mem;
#endif
  }
  ELL_HD static bool writer() { return CoopK256::writer(); }
  ELL_HD static A load_entry(const typename W1::A* e) {
    A q;
    q.x = F::load_words(e->x.v);
    q.y = F::load_words(e->y.v);
    return q;
  }
  ELL_HD static A load_affine(const u8* xy, size_t i) {
    u32 tx[L], ty[L];
    load_be<L>(tx, xy + i * 2 * W1::BYTES, W1::BYTES);
    load_be<L>(ty, xy + i * 2 * W1::BYTES + W1::BYTES, W1::BYTES);
    A q;
    q.x = F::from_plain(tx);
    q.y = F::from_plain(ty);
    return q;
  }
  // -> the one-lane kernels' format (W1::store_jac: the field's words, limb-major; FpSolinas: plain)
  ELL_HD static void store_jac(u32* jac, size_t n, size_t i, const J& p) {
    u32 x[L], y[L], z[L];
    F::to_plain(x, p.X);
    F::to_plain(y, p.Y);
    F::to_plain(z, p.Z);
    if (CoopK256::writer()) {
      ELL_UNROLL
      for (int l = 0; l < L; l++) {
        jac[(size_t)(0 * L + l) * n + i] = x[l];
        jac[(size_t)(1 * L + l) * n + i] = y[l];
        jac[(size_t)(2 * L + l) * n + i] = z[l];
      }
    }
  }
  // the odd multiples of P on the curve itself (co-Z table on the isomorphic curves, one inversion
  // to map it back to the curve whose a = -3 doubling the ladder needs): the first half of var_ladder
  ELL_HD static void build_table(A* tbl, const A& p) {
    El zg;
    LD::template build_table_odd8<NE>(tbl, p, zg);
    El zi = F::inv(zg);
    El zi2 = F::sqr(zi);
    El zi3 = F::mul(zi2, zi);
    ELL_NOUNROLL
    for (int e = 0; e < NE; e++) {
      A t = tbl[e];
      t.x = F::mul(t.x, zi2);
      t.y = F::mul(t.y, zi3);
      tbl[e] = t;
    }
  }
  // the digits of k over such a table (k even -> k + 1, P subtracted at the end)
  ELL_HD static J run_table(const u32 (&k)[L], const DigitStore& ds, const A* tbl) {
    u32 kk[L];
    bn_copy<L>(kk, k);
    const u32 evenmask = (k[0] & 1u) ? 0u : 1u;
    kk[0] |= 1u;
    recode_odd_w4<L, NW, WB>(kk, ds, 0, 1);
    bool inf;
    return LD::template run_odd_w4<1, NW, false, false, WB>(ds, tbl, 0u, evenmask, inf);
  }
  // k*P (work.h var_ladder, the curves without an endomorphism)
  ELL_HD static J ladder(const u32 (&k)[L], const A& p, const DigitStore& ds, A* tbl) {
    build_table(tbl, p);
    return run_table(k, ds, tbl);
  }
  // A unit's table in global memory, in the ROW's own format (nothing is converted): the table of an
  // EC#verify's key does not depend on s^-1, so a unit of the launch in front builds it beside the
  // scalar-field prep (FnEcdsaPrepTableN) and the ladder's unit starts from it.  GROW lane-entries
  // per table entry (the device keeps a limb per lane, host passes a whole row per element).
  static constexpr int GROW = F::CL == 1 ? LANES : 1;
  static constexpr size_t TABLE_BYTES = (size_t)NE * GROW * sizeof(A);
  ELL_HD static void table_out(A* g, const A* tbl) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (threadIdx.x < (unsigned)LANES) {
      ELL_NOUNROLL
      for (int e = 0; e < NE; e++) g[e * LANES + threadIdx.x] = tbl[e];
    }
#else
    for (int e = 0; e < NE; e++) g[e] = tbl[e];
#endif
  }
  ELL_HD static void table_in(A* tbl, const A* g) {
#if defined(__HIP_DEVICE_COMPILE__)
    ELL_NOUNROLL
    for (int e = 0; e < NE; e++) tbl[e] = g[e * LANES + (threadIdx.x & (unsigned)(LANES - 1))];
#else
    for (int e = 0; e < NE; e++) tbl[e] = g[e];
#endif
  }
  ELL_HD static void ecdsa_table(size_t i, const u8* pub_xy, A* gtbl, void* row_mem) {
    A* tbl = lane_table(row_mem);
    build_table(tbl, load_affine(pub_xy, i));
    table_out(gtbl + i * (size_t)NE * GROW, tbl);
  }
  // EC#verify part 0: u2*Q
  // (gtbl != null: over the table ecdsa_table built in the launch in front)
  ELL_HD static void ecdsa_var(size_t i, size_t n, const u32* u12, const u8* pub_xy, const DigitStore& ds,
                               u32* jac, void* row_mem, const A* gtbl) {
    u32 u2[L];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u2[l] = l < W1::LN ? u12[(size_t)(1 * W1::LN + l) * n + i] : 0u;
    A* tbl = lane_table(row_mem);
    if (gtbl) {
      table_in(tbl, gtbl + i * (size_t)NE * GROW);
      store_jac(jac, n, i, run_table(u2, ds, tbl));
    } else {
      store_jac(jac, n, i, ladder(u2, load_affine(pub_xy, i), ds, tbl));
    }
  }
  // EC#verify part 1: u1*G
  ELL_HD static void ecdsa_fixed(size_t i, size_t n, const u32* u12, const typename W1::A* comb, u32* jac) {
    u32 u1[L];
    ELL_UNROLL
    for (int l = 0; l < L; l++) u1[l] = l < W1::LN ? u12[(size_t)(0 * W1::LN + l) * n + i] : 0u;
    store_jac(jac, n, i, coop_comb_mul<CoopNist<CV1>>(u1, comb));
  }
  // Point#mul
  ELL_HD static void mul_var(size_t i, size_t n, const u8* ks, const u8* xy, const DigitStore& ds, u32* jac,
                             void* row_mem) {
    u32 k[L];
    load_be<L>(k, ks + i * W1::BYTES, W1::BYTES);
    store_jac(jac, n, i, ladder(k, load_affine(xy, i), ds, lane_table(row_mem)));
  }
  // the comb of k1 in k1*G + k2*P
  ELL_HD static void mul_fixed_part(size_t i, size_t n, const u8* ks, const typename W1::A* comb, u32* jac) {
    u32 k[L];
    load_be<L>(k, ks + i * W1::BYTES, W1::BYTES);
    store_jac(jac, n, i, coop_comb_mul<CoopNist<CV1>>(k, comb));
  }
};

}  // namespace ell
