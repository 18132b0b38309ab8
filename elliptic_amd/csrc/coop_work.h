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

namespace ell {

struct CoopK256 {
  typedef CvSecp256k1C CV;
  typedef FpK256C F;
  typedef F::El El;
  typedef ShortOps<CV> G;
  typedef Ladder<CV> LD;
  typedef Jac<F> J;
  typedef Aff<F> A;
  typedef Work<CvSecp256k1> W1;                      // the one-lane layer: geometry, table and result formats
  typedef W1::Endo<true> E;                          // the small-grid tuning's windows (5 bits)
  // the row layer reads and writes the SATURATED field's tables and results; the ELL_K256_LAZY
  // build (one-lane kernels on fpk256l.h) keeps its small batches on the one-lane parts
  static constexpr bool AVAILABLE = std::is_same<CvSecp256k1::F, FpK256>::value;
  // table slots of one lane: the odd multiples + build_table_odd8's ratios
  static constexpr int SLOTS = 2 * E::NE;
  // bytes of row memory a unit needs (k_run_coop's LDS / the host loop's stack): SLOTS entries of
  // every lane of the row
  static constexpr int ROW_BYTES = SLOTS * FpK256C::ROW * 16;
  static_assert(sizeof(A) * (16 / FpK256C::CL) <= 256 && sizeof(A) * SLOTS * (FpK256C::CL == 1 ? 16 : 1) <= (size_t)ROW_BYTES, "row memory too small");

  // this lane's table: SLOTS entries of its own
  ELL_HD static A* lane_table(void* row_mem) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (A*)row_mem + (size_t)(threadIdx.x & 15u) * SLOTS;
#else
    return (A*)row_mem;
#endif
  }
  // does this thread store the (wave-uniform) results?
  ELL_HD static bool writer() {
#if defined(__HIP_DEVICE_COMPILE__)
    return threadIdx.x == 0;
#else
    return true;
#endif
  }

  // a point in the one-lane kernels' memory format (Aff<FpK256>: x, y as eight plain words) -> the row
  ELL_HD static A load_entry(const W1::A* e) {
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
  ELL_HD static void ecdsa_half(size_t i, size_t n, int half, const u32* u12, const W1::VT* tbl_all,
                                const DigitStore& ds, u32* jac, void* row_mem) {
    u32 u2[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++) u2[l] = u12[(size_t)(1 * W1::LN + l) * n + i];
    const bool lam = half != 0;
    const u32 negmask = half_digits(u2, lam, ds) ? 1u : 0u;
    A* tbl = lane_table(row_mem);
    const W1::VT* src = tbl_all + i * W1::stride<true>();
    ELL_NOUNROLL
    for (int e = 0; e < E::NE; e++) tbl[e] = load_entry(src + e);
    El beta = load_beta();
    bool inf;
    J b = LD::template run_odd_w4<1, E::NW, true, false, E::WB>(ds, tbl, negmask, 0u, inf, &beta, lam);
    store_jac(jac, n, i, b);
  }

  // ---- fixed base: k*G over the one-lane comb table (ladder.h comb_add, entries through load_entry) ----
  ELL_HD static J comb_mul(const u32 (&k)[8], const W1::A* comb) {
    constexpr bool SIGNED = W1::COMB_SIGNED;
    const int CB = SIGNED ? comb_bits_of(comb) : W1::COMB_BITS;        // (the table carries its window width)
    const int W = SIGNED ? comb_windows(256, CB) : W1::COMB_W;
    const u32 MASK = (1u << CB) - 1u;
    const u32 HALF = 1u << (CB - 1);
    const u32 PER = SIGNED ? HALF : MASK;
    u32 kk[8];
    bn_copy<8>(kk, k);
    u32 carry = 0;
    J acc = G::infinity();
    bool inf = true;
    ELL_NOUNROLL
    for (int w = 0; w < W; w++) {
      u32 d = (kk[0] & MASK) + carry;
      ELL_UNROLL
      for (int j = 0; j < 7; j++) kk[j] = (kk[j] >> CB) | (kk[j + 1] << (32 - CB));
      kk[7] >>= CB;
      bool neg = false;
      if (SIGNED) {
        neg = d > HALF;
        carry = neg ? 1u : 0u;
        d = neg ? (MASK + 1u) - d : d;
      }
      if (d != 0) {                                      // wave-uniform: one item per wave
        const W1::A* e = comb + ((size_t)w * PER + (d - 1u));
        auto fetch = [&]() -> A {
          A q = load_entry(e);
          if (SIGNED) q.y = LD::cneg_y(q.y, neg);
          return q;
        };
        acc = G::add_mixed_lean(acc, fetch(), inf, fetch);
      }
    }
    return acc;
  }
  // EC#verify, part 2: u1*G
  ELL_HD static void ecdsa_fixed(size_t i, size_t n, const u32* u12, const W1::A* comb, u32* jac) {
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
  ELL_HD static void mul_fixed_part(size_t i, size_t n, const u8* ks, const W1::A* comb, u32* jac) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    store_jac(jac, n, i, comb_mul(k, comb));
  }
};

}  // namespace ell
