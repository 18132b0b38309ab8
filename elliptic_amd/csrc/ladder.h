// ellgpu -- scalar recoding, per-lane window tables, ladders and the
// fixed-base comb for short Weierstrass curves.
//
// What the reference does with data-dependent wNAF / JSF digit strings
// (lib/elliptic/curve/base.js:52-253, utils.js:15-101) is done here with
// REGULAR recodings so that the 64 lanes of a wavefront -- each working on its
// own (scalar, point) -- execute the same add/double sequence:
//
//   variable base  signed fixed 4-bit windows (digits -8..7 via a bias, so no
//                  carry chain), 8-entry table {1..8}P per lane in HBM/L2
//                  scratch, one add per window, every lane adds at the same
//                  step (a zero digit is a select, not a branch);
//   secp256k1      GLV first (k = k1 + k2*lambda, |k1|,|k2| < 2^129), two
//                  33-window ladders sharing the doublings
//                  (replaces _endoSplit/_endoWnafMulAdd, short.js:168-249);
//   fixed base     unsigned 8-bit comb over a table of d*2^(8w)*G, no
//                  doublings at all (replaces _fixedNafMul, base.js:52-84).
//
// Group results are independent of the recoding, so outputs stay bit-exact.
#pragma once

#include "curves.h"
#include "short.h"

namespace ell {

// Per-lane digit strings.  On the GPU `base` points into LDS at the lane's
// column and `stride` is the workgroup size (bank-conflict-free byte
// columns); in the host simulation it is a plain array with stride 1.
struct DigitStore {
  signed char* base;
  int stride;
  ELL_HD void set(int idx, int d) const { base[idx * stride] = (signed char)d; }
  ELL_HD int get(int idx) const { return base[idx * stride]; }
};

// number of limbs needed for NNIB nibbles plus one carry bit
template <int NNIB>
struct RecodeLimbs {
  static constexpr int LK = (NNIB * 4) / 32 + 1;
};

// Signed 4-bit recoding of k < 16^NNIB into windows [0, NNIB) with digits
// -8..7, plus (TOP) one more window holding the 0/1 carry:
//   k + sum_i 8*16^i = sum_i nib_i 16^i   =>   k = sum_i (nib_i - 8) 16^i (+ top*16^NNIB)
// Digits are written to ds at index w*NS + s.
template <int LW, int NNIB, bool TOP>
ELL_HD void recode_w4(const u32 (&k)[LW], const DigitStore& ds, int s, int NS) {
  constexpr int LK = RecodeLimbs<NNIB>::LK;
  u32 kp[LK];
  u64 c = 0;
  ELL_UNROLL
  for (int i = 0; i < LK; i++) {
    // bias limb i: nibbles 8 for nibble index < NNIB
    u32 b = 0;
    ELL_UNROLL
    for (int j = 0; j < 8; j++)
      if (8 * i + j < NNIB) b |= 8u << (4 * j);
    c += (u64)(i < LW ? k[i] : 0u) + b;
    kp[i] = (u32)c;
    c >>= 32;
  }
  ELL_UNROLL
  for (int i = 0; i < NNIB; i++) {
    int nib = (int)((kp[i >> 3] >> (4 * (i & 7))) & 15u);
    ds.set(i * NS + s, nib - 8);
  }
  if (TOP) {
    int nib = (int)((kp[NNIB >> 3] >> (4 * (NNIB & 7))) & 15u);
    ds.set(NNIB * NS + s, nib);
  }
}

// Signed ODD WB-bit recoding of an odd k < 2^(WB*NW): with the bit 2^(WB*NW) forced on,
//   k = sum_i d_i 2^(WB i),  d_i = 2*((k' >> (WB i + 1)) & (2^WB - 1)) - (2^WB - 1)  in {+-1, +-3, ..., +-(2^WB - 1)}
// (every window non-zero, so the ladder adds at every step and needs only the 2^(WB-1) odd
// multiples).  Digits go to ds at index w*NS + s.
template <int LW, int NW, int WB = 4>
ELL_HD void recode_odd_w4(const u32 (&k)[LW], const DigitStore& ds, int s, int NS) {
  constexpr int LK = (WB * NW) / 32 + 1;
  constexpr u32 DM = (1u << WB) - 1u;
  u32 kp[LK + 1];
  ELL_UNROLL
  for (int i = 0; i < LK + 1; i++) kp[i] = i < LW ? k[i] : 0u;
  kp[(WB * NW) >> 5] |= 1u << ((WB * NW) & 31);
  ELL_UNROLL
  for (int i = 0; i < NW; i++) {
    const int bit = WB * i + 1;
    const int li = bit >> 5, sh = bit & 31;
    u32 v = kp[li] >> sh;
    if (sh > 32 - WB) v |= kp[li + 1] << (32 - sh);
    ds.set(i * NS + s, 2 * (int)(v & DM) - (int)DM);
  }
}

// window width of a signed fixed-base table (see Ladder::comb_add): the word in front of entry 0
template <class AA>
ELL_HD int comb_bits_of(const AA* comb) { return (int)((const u32*)(comb - 1))[0]; }
// windows of a signed cb-bit comb over `bits`-bit scalars (one more bit for the recoding's carry)
ELL_HD constexpr int comb_windows(int bits, int cb) { return (bits + cb) / cb; }

template <class CV>
struct Ladder {
  typedef typename CV::F F;
  typedef typename F::El El;
  typedef ShortOps<CV> G;
  typedef Jac<F> J;
  typedef Aff<F> A;

  // beta for the lambda-at-lookup entries.  ELL_BETA_REMAT = 1: instead of holding the
  // eight limbs in VGPRs across the whole ladder, move them in from scalar registers at every
  // lookup (8 v_mov per lambda*P lookup, 8 registers fewer live in the loop).  WIDE (the
  // register-rich small-grid build, common.h) keeps them in VGPRs.
  template <bool L, bool WIDE = false>
  ELL_HD static El lookup_beta(const El* beta) {
#if ELL_BETA_REMAT && defined(__HIP_DEVICE_COMPILE__)
    if constexpr (L && !WIDE && !is_lazy<F>::value) {
      El b;
      ELL_UNROLL
      for (int i = 0; i < F::L; i++) {
        u32 c;
        asm volatile("v_mov_b32 %0, %1" : "=v"(c) : "s"(CV::C::beta[i]));
        b.v[i] = c;
      }
      return b;
    }
#endif
    return *beta;
  }
  // neg ? -y : y for a table entry's y (a direct product output): lazy fields negate limb-wise
  // with the offset 2p, without a normalisation pass
  ELL_HD static El cneg_y(const El& y, bool neg) {
    if constexpr (is_lazy<F>::value) return F::template cneg_l<2>(y, neg);
    else return fe_select<F>(neg, F::neg(y), y);
  }

  // tbl[j-1] = j*P for j = 1..8 (Jacobian, distinct Z)
  ELL_HD static void build_table8(J* tbl, const A& p) {
    tbl[0] = G::from_affine(p);
    ELL_NOUNROLL
    for (int j = 2; j <= 8; j++) {
      J t;
      if (j & 1) t = G::add_mixed(tbl[j - 2], p);
      else t = G::dbl(tbl[j / 2 - 1]);
      tbl[j - 1] = t;
    }
  }

  // Effective-affine table of the odd multiples {1,3,...,15}*P:
  // with d = 2P = (Xd, Yd, Zd), the map (x, y) -> (x Zd^2, y Zd^3) sends the curve to an
  // isomorphic one (a' = a Zd^4; additions do not involve a) on which d is AFFINE, so each next odd multiple is
  // one mixed add; rescaling every entry to the last entry's Z then makes the whole table
  // affine on a second isomorphic curve.  The ladder runs there with 8M+3S mixed adds
  // (instead of 12M+4S) and its result is mapped back by one multiplication of Z by
  // zg = Z_last * Zd -- when a = 0, whose doubling formula holds on every isomorphic curve;
  // for a = -3 the caller maps the TABLE back to the true curve with zg^-1 instead
  // (Work::var_ladder).   tbl[0..8) = table, tbl[8..16) is used as scratch for the ratios.
  template <int NE = 8>
  ELL_HD static void build_table_odd8(A* tbl, const A& p, El& zg) {
    J d = G::dbl(G::from_affine(p));
    El zd2 = F::sqr(d.Z);
    El zd3 = F::mul(zd2, d.Z);
    A dd; dd.x = d.X; dd.y = d.Y;
    J t;
    t.X = F::mul(p.x, zd2);
    t.Y = F::mul(p.y, zd3);
    t.Z = F::one();
    tbl[0].x = t.X; tbl[0].y = t.Y;
    if constexpr (is_lazy<F>::value || !ELL_COZ_TABLE) {
      ELL_NOUNROLL
      for (int i = 1; i < NE; i++) {
        El h;
        t = G::add_mixed_zr(t, dd, h);
        tbl[i].x = t.X; tbl[i].y = t.Y;
        tbl[NE + i].x = h;
      }
    } else {
      // Co-Z chain (Meloni's ZADDU): on the first isomorphic curve P and 2P are both affine, i.e.
      // share Z = 1; an addition of two points with a COMMON Z costs 4M + 2S here -- the product
      // Z3 = Z h is never formed, only its factor h = X_2P - X_jP is kept for the rescaling pass --
      // and hands back 2P on the new Z, ready for the next step: 4M + 2S per odd multiple instead
      // of the mixed addition's 8M + 3S.
      El x2 = t.X, y2 = t.Y;                               // the current odd multiple j P
      ELL_NOUNROLL
      for (int i = 1; i < NE; i++) {
        El h = F::sub(dd.x, x2);
        El c = F::sqr(h);
        El w1 = F::mul(dd.x, c);
        El w2 = F::mul(x2, c);
        El dy = F::sub(dd.y, y2);
        El dsq = F::sqr(dy);
        El a1 = F::mul(dd.y, F::sub(w1, w2));
        El x3 = F::sub(F::sub(dsq, w1), w2);
        El y3 = F::sub(F::mul(dy, F::sub(w1, x3)), a1);
        dd.x = w1; dd.y = a1;                              // 2P on the new Z
        x2 = x3; y2 = y3;                                  // (j + 2) P
        tbl[i].x = x3; tbl[i].y = y3;
        tbl[NE + i].x = h;
      }
    }
    // every entry to the last entry's Z: zr = h_(i+1) ... h_7, x zr^2, y zr^3; entry 0 had Z = 1,
    // so its ratio is the chain's final Z
    El zr = F::one();
    ELL_NOUNROLL
    for (int i = NE - 2; i >= 0; i--) {
      zr = F::mul(zr, tbl[NE + i + 1].x);
      El zr2 = F::sqr(zr);
      A e = tbl[i];
      e.x = F::mul(e.x, zr2);
      e.y = F::mul(e.y, F::mul(zr2, zr));
      tbl[i] = e;
    }
    zg = F::mul(zr, d.Z);
  }

  // acc = sum_s k_s * P_s on the effective-affine curve: digits odd (recode_odd_w4), tables
  // tbl[s*8 + (|d|-1)/2] affine, point s negated when bit s of negmask is set; afterwards
  // P_s is subtracted once where bit s of evenmask is set (k_s had been made odd by +1).
  // LAMBDA_AT_LOOKUP (secp256k1, NS = 2): the second digit string's table (lambda * P = (beta x, y))
  // is not stored; its entries are the first table's with x multiplied by beta at every lookup.
  // Halves the table bytes written per item and the region the gathers touch, for one more
  // field multiplication per addition of the second string (ELL_LAMBDA_AT_LOOKUP).
  template <int NS, int NW, bool LAMBDA_AT_LOOKUP = false, bool WIDE = false, int WB = 4>
  // `lam_all` (NS = 1, the parted ladder of Work::ecdsa_half): this lane's ONE digit string is the
  // lambda half's -- every entry it looks up takes the factor beta.
  ELL_HD static J run_odd_w4(const DigitStore& ds, const A* tbl, u32 negmask, u32 evenmask, bool& inf,
                             const El* beta = nullptr, bool lam_all = false) {
    // table entry for digit string s at window w (digits are odd and non-zero); a function of
    // (w, s) only, so that the additions' rarely taken branch can fetch it again
    auto entry = [&](int w, int s) -> A {
      int d = ds.get(w * NS + s);
      int ad = d < 0 ? -d : d;
      bool neg = (d < 0) != (((negmask >> s) & 1u) != 0);
      A q = tbl[(LAMBDA_AT_LOOKUP ? 0 : s * (1 << (WB - 1))) + ((ad - 1) >> 1)];
      if (LAMBDA_AT_LOOKUP && (s == 1 || lam_all)) q.x = F::mul(q.x, lookup_beta<LAMBDA_AT_LOOKUP, WIDE>(beta));
      q.y = cneg_y(q.y, neg);
      return q;
    };
    // top window, first string: acc = the entry itself (no addition into O)
    J acc = G::from_affine(entry(NW - 1, 0));
    inf = false;
    if constexpr (WIDE || ELL_PREFETCH) {
    // Software-pipelined form (the register-rich small-grid build): the table entry of the next
    // addition is requested before the work that precedes that addition -- the first string's
    // entry before the window's four doublings, the second string's before the first addition --
    // so that at two resident waves per SIMD no gather latency is exposed.  `raw` = the stored
    // entry; sign and beta are applied at the point of use.
    auto raw = [&](int w, int s, bool& neg, const A*& at) -> A {
      int d = ds.get(w * NS + s);
      int ad = d < 0 ? -d : d;
      neg = (d < 0) != (((negmask >> s) & 1u) != 0);
      at = tbl + (LAMBDA_AT_LOOKUP ? 0 : s * (1 << (WB - 1))) + ((ad - 1) >> 1);
      return *at;
    };
    auto finish = [&](A q, int s, bool neg) -> A {
      if (LAMBDA_AT_LOOKUP && (s == 1 || lam_all)) q.x = F::mul(q.x, lookup_beta<LAMBDA_AT_LOOKUP, WIDE>(beta));
      q.y = cneg_y(q.y, neg);
      return q;
    };
    static_assert(NS <= 2, "prefetch form written for one or two digit strings");
    ELL_NOUNROLL
    for (int w = NW - 1; w >= 0; w--) {
      bool neg0 = false, neg1 = false;
      const A* at0 = tbl;
      const A* at1 = tbl;
      A q0, q1;
      if (w != NW - 1) {
        q0 = raw(w, 0, neg0, at0);
        ELL_NOUNROLL
        for (int j = 0; j < WB; j++) acc = G::dbl(acc);
      }
      if (NS > 1) q1 = raw(w, NS - 1, neg1, at1);
      if (w != NW - 1)
        acc = G::add_mixed_lean(acc, finish(q0, 0, neg0), inf, [&]() { return finish(*at0, 0, neg0); });
      if (NS > 1)
        acc = G::add_mixed_lean(acc, finish(q1, NS - 1, neg1), inf, [&]() { return finish(*at1, NS - 1, neg1); });
    }
    } else {
    ELL_NOUNROLL
    for (int w = NW - 1; w >= 0; w--) {
      if (w != NW - 1) {
        ELL_NOUNROLL
        for (int j = 0; j < WB; j++) acc = G::dbl(acc);
      }
      ELL_NOUNROLL
      for (int s = (w == NW - 1 ? 1 : 0); s < NS; s++)
        acc = G::add_mixed_lean(acc, entry(w, s), inf, [&]() { return entry(w, s); });
    }
    }
    ELL_NOUNROLL
    for (int s = 0; s < NS; s++) {
      auto corr = [&]() -> A {
        A q = tbl[LAMBDA_AT_LOOKUP ? 0 : s * (1 << (WB - 1))];
        if (LAMBDA_AT_LOOKUP && (s == 1 || lam_all)) q.x = F::mul(q.x, lookup_beta<LAMBDA_AT_LOOKUP, WIDE>(beta));
        bool neg = ((negmask >> s) & 1u) == 0;        // subtract sign_s * P_s
        q.y = cneg_y(q.y, neg);
        return q;
      };
      // per-lane condition: lanes without the correction sit the addition out (exec mask)
      if (((evenmask >> s) & 1u) != 0) acc = G::add_mixed_lean(acc, corr(), inf, corr);
    }
    return acc;
  }

  // acc = sum_s k_s * P_s, digits from ds (NWIN windows of NS digits), tables
  // tbl[s*8 + (|d|-1)], point s negated when bit s of negmask is set.
  template <int NS, int NWIN>
  ELL_HD static J run_w4(const DigitStore& ds, const J* tbl, u32 negmask) {
    J acc = G::infinity();
    ELL_NOUNROLL
    for (int w = NWIN - 1; w >= 0; w--) {
      if (w != NWIN - 1) {
        ELL_NOUNROLL
        for (int j = 0; j < 4; j++) acc = G::dbl(acc);
      }
      ELL_NOUNROLL
      for (int s = 0; s < NS; s++) {
        int d = ds.get(w * NS + s);
        int ad = d < 0 ? -d : d;
        bool neg = (d < 0) != (((negmask >> s) & 1u) != 0);
        int e = ad ? ad - 1 : 0;
        J q = tbl[s * 8 + e];
        q = G::cneg(q, neg);
        acc = G::add(acc, q, ad != 0);
      }
    }
    return acc;
  }

  // fixed-base comb: acc + sum_w d_w * 2^(CB*w) * G with d_w the w-th CB-bit digit of k;
  // UNSIGNED (CB = 8): comb[w*(2^CB - 1) + d-1] = d * 2^(CB*w) * G;  SIGNED (the 256-bit curves,
  // CB = 22): digits recoded on the fly into [-2^(CB-1), 2^(CB-1)] (a window above 2^(CB-1) takes
  // its value minus 2^CB and carries one into the next), comb[w*2^(CB-1) + |d|-1] = |d| * 2^(CB*w) * G
  // and y negated at lookup -- half the table per window bit, so wider windows for the same bytes.
  // Entries are affine, field-internal form.  Zero digits sit the addition out (exec mask).
  // `inf` = acc is O, updated.  WIDE: window w+1's entry is requested before window w's addition.
  // The SIGNED comb's window width is a property of the TABLE, not of the kernel: the slot in front
  // of entry 0 holds it (comb_bits_of), so that Engine::ensure_comb can narrow the comb when the
  // device cannot hold the default 1.6 GB one (ELLGPU_E_NOMEM otherwise: ADVICE r3) -- the same
  // kernels then run 17 / 33 / 65 windows of 16 / 8 / 4 bits instead of 12 of 22.  CB / W are the
  // default geometry (and the unsigned combs' only one).
  template <int LK, int W, int CB, bool WIDE = false, bool SIGNED = false>
  ELL_HD static J comb_add(J acc, bool& inf, const u32 (&k)[LK], const A* comb) {
    const int cb = SIGNED ? comb_bits_of(comb) : CB;
    const int nwin = SIGNED ? (LK * 32 + cb) / cb : W;    // signed: one more bit for the recoding's carry
    const u32 MASK = (1u << cb) - 1u;
    const u32 HALF = 1u << (cb - 1);
    const u32 PER = SIGNED ? HALF : MASK;                 // entries per window
    u32 kk[LK];
    bn_copy<LK>(kk, k);
    u32 carry = 0;
    // next digit -> (table index within the window, negate flag); d == 0 -> idx 0 with zero = true
    auto digit = [&](u32& idx, bool& neg, bool& zero) {
      u32 d = (kk[0] & MASK) + carry;
      ELL_UNROLL
      for (int i = 0; i < LK - 1; i++) kk[i] = (kk[i] >> cb) | (kk[i + 1] << (32 - cb));
      kk[LK - 1] >>= cb;
      if (SIGNED) {
        neg = d > HALF;
        carry = neg ? 1u : 0u;
        d = neg ? (MASK + 1u) - d : d;                     // |d| in [0, 2^(CB-1)]
      } else {
        neg = false;
      }
      zero = d == 0;
      idx = zero ? 0u : d - 1u;
    };
    auto fetch = [&](const A* e, bool neg) -> A {
      A q = *e;
      if (SIGNED) q.y = cneg_y(q.y, neg);
      return q;
    };
    if constexpr (WIDE || ELL_PREFETCH) {
      u32 idx; bool neg, zero;
      digit(idx, neg, zero);
      const A* e = comb + idx;
      A q = *e;
      ELL_NOUNROLL
      for (int w = 0; w < nwin; w++) {
        u32 idxn = 0; bool negn = false, zeron = true;
        const A* en = comb;
        A qn = q;
        if (w + 1 < nwin) {
          digit(idxn, negn, zeron);
          en = comb + ((size_t)(w + 1) * PER + idxn);
          qn = *en;
        }
        if (!zero) {
          if (SIGNED) q.y = cneg_y(q.y, neg);
          acc = G::add_mixed_lean(acc, q, inf, [&]() { return fetch(e, neg); });
        }
        idx = idxn; neg = negn; zero = zeron; e = en; q = qn;
      }
      return acc;
    } else {
      ELL_NOUNROLL
      for (int w = 0; w < nwin; w++) {
        u32 idx; bool neg, zero;
        digit(idx, neg, zero);
        if (!zero) {
          const A* e = comb + ((size_t)w * PER + idx);
          acc = G::add_mixed_lean(acc, fetch(e, neg), inf, [&]() { return fetch(e, neg); });
        }
      }
      return acc;
    }
  }
  template <int LK, int W, int CB, bool SIGNED = false>
  ELL_HD static J comb_mul(const u32 (&k)[LK], const A* comb) {
    bool inf = true;
    return comb_add<LK, W, CB, false, SIGNED>(G::infinity(), inf, k, comb);
  }
};

// --------------------------------------------------------------------------
// GLV decomposition for secp256k1 (replaces ShortCurve#_endoSplit,
// lib/elliptic/curve/short.js:168-185).  c1 = round(b2*k/n), c2 =
// round(-b1*k/n) are taken as the top 128 bits of k*g with g = round(2^384 *
// b/n); an off-by-one in c only moves (k1,k2) by one lattice vector, so
// k1 + k2*lambda == k (mod n) holds exactly and |k1|,|k2| < 2^129.
// --------------------------------------------------------------------------
ELL_HD void glv_mul_shift384(u32 (&c)[4], const u32 (&k)[8], const u32 (&g)[8]) {
  u32 t[16];
  bn_mul_wide<8, 8>(t, k, g);
  // + 2^383, then >> 384
  u64 x = (u64)t[11] + 0x80000000u;
  u32 cy = (u32)(x >> 32);
  ELL_UNROLL
  for (int i = 0; i < 4; i++) {
    u64 y = (u64)t[12 + i] + cy;
    c[i] = (u32)y;
    cy = (u32)(y >> 32);
  }
}

// |x| for a 9-limb two's complement value; returns sign
ELL_HD bool abs9(u32 (&r)[9]) {
  bool neg = (r[8] >> 31) != 0;
  u32 m = neg ? 0xFFFFFFFFu : 0u;
  u64 c = neg ? 1 : 0;
  ELL_UNROLL
  for (int i = 0; i < 9; i++) {
    c += (u64)(r[i] ^ m);
    r[i] = (u32)c;
    c >>= 32;
  }
  return neg;
}

// ODD = true: both halves come back odd.  (k1, k2) may be moved by any lattice vector
// (a_i, b_i) -- a_i + b_i * lambda == 0 (mod n) -- and the basis' parities, (a1, b1) = (1, 1) and
// (a2, b2) = (0, 1) mod 2, span all four cases: one conditional add / subtract of each vector fixes
// both parities, |k1|, |k2| stay below 2^130 (the 33-window odd recoding takes < 2^132), and the
// ladder needs no "k was even: subtract P once more" additions at its end (two mixed additions
// that every wave executed, 1 % of a verify).  Signs are chosen towards zero.
template <bool ODD = false>
ELL_HD void glv_split(const u32 (&k)[8], u32 (&k1)[5], bool& neg1, u32 (&k2)[5], bool& neg2) {
  typedef consts::SECP256K1_C C;
  u32 g1[8], g2[8], a1[4], mb1[4], a2[5], b2[4];
  ELL_UNROLL
  for (int i = 0; i < 8; i++) { g1[i] = C::glv_g1[i]; g2[i] = C::glv_g2[i]; }
  ELL_UNROLL
  for (int i = 0; i < 4; i++) { a1[i] = C::glv_a1[i]; mb1[i] = C::glv_mb1[i]; b2[i] = C::glv_b2[i]; }
  ELL_UNROLL
  for (int i = 0; i < 5; i++) a2[i] = C::glv_a2[i];
  u32 c1[4], c2[4];
  glv_mul_shift384(c1, k, g1);
  glv_mul_shift384(c2, k, g2);
  // k1 = k - c1*a1 - c2*a2
  u32 p1[8], p2[9];
  bn_mul_wide<4, 4>(p1, c1, a1);
  bn_mul_wide<4, 5>(p2, c2, a2);
  u32 r[9];
  {
    u32 br = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      u64 t = (u64)(i < 8 ? k[i] : 0u) - (i < 8 ? p1[i] : 0u) - br;
      r[i] = (u32)t;
      br = (u32)(t >> 63);
    }
    br = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      u64 t = (u64)r[i] - p2[i] - br;
      r[i] = (u32)t;
      br = (u32)(t >> 63);
    }
  }
  // k2 = -(c1*b1 + c2*b2) = c1*|b1| - c2*b2
  u32 q1[8], q2[8], r2[9];
  bn_mul_wide<4, 4>(q1, c1, mb1);
  bn_mul_wide<4, 4>(q2, c2, b2);
  {
    u32 br = 0;
    ELL_UNROLL
    for (int i = 0; i < 9; i++) {
      u64 t = (u64)(i < 8 ? q1[i] : 0u) - (i < 8 ? q2[i] : 0u) - br;
      r2[i] = (u32)t;
      br = (u32)(t >> 63);
    }
  }
  if constexpr (ODD) {
    // 9-limb two's complement x += sign * v (v >= 0, nv limbs), sign = +1 / -1, only where `on`
    auto addv = [](u32 (&x)[9], const u32* v, int nv, bool minus, bool on) {
      u64 c = (on && minus) ? 1 : 0;                      // x - v = x + ~v + 1 over all nine limbs
      ELL_UNROLL
      for (int i = 0; i < 9; i++) {
        const u32 vi = i < nv ? v[i] : 0u;
        const u32 w = on ? (minus ? ~vi : vi) : 0u;
        c += (u64)x[i] + w;
        x[i] = (u32)c;
        c >>= 32;
      }
    };
    // v1 = (a1, -|b1|): fixes k1's parity; towards zero in its large component, k2
    const bool x1 = (r[0] & 1u) == 0;
    const bool k2pos = (r2[8] >> 31) == 0;
    addv(r, a1, 4, !k2pos, x1);                           // k2 > 0: + v1 (k2 shrinks by |b1|), else - v1
    addv(r2, mb1, 4, k2pos, x1);
    // v2 = (a2, b2): k2's parity (a2 is even); towards zero in its large component, k1
    const bool x2 = (r2[0] & 1u) == 0;
    const bool k1pos = (r[8] >> 31) == 0;
    addv(r, a2, 5, k1pos, x2);                            // k1 > 0: - v2, else + v2
    addv(r2, b2, 4, k1pos, x2);
  }
  neg1 = abs9(r);
  ELL_UNROLL
  for (int i = 0; i < 5; i++) k1[i] = r[i];
  neg2 = abs9(r2);
  ELL_UNROLL
  for (int i = 0; i < 5; i++) k2[i] = r2[i];
}

}  // namespace ell
