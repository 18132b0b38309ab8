// ellgpu -- ed25519 on the lanes-per-item layer (coop.h): EDDSA#verify for a handful of items.
//
// The reference's verify is one chain: decode A and R (a square root each), h*A by the window
// ladder, S*G, one addition, one comparison (eddsa/index.js:52-63, curve/edwards.js:174-205,
// 279-309, 362-375).  Here an item is TWO waves -- part 0: h = SHA-512(R || A || M), A, its window
// table and h*A; part 1: R, S*G over the one-lane comb table, S*G - R -- and the one-lane
// eddsa_join compares the two projective points.  Inside a wave a field element is one DPP row
// (nine signed 29-bit limbs, coop.h), and the extended-coordinate formulas are what the four rows
// of a wave are made for: a doubling is two steps of FOUR independent products (X^2, Y^2, Z^2,
// (X+Y)^2, then E F, G H, E H, F G), an addition likewise (coop.h Q / mulq).
//
// The field: signed limbs and the value fold of coop_mont.h (norm: one estimate from the top limb,
// no offsets), but no Montgomery form -- 2^261 = 1216 (mod 2^255 - 19), so a product's columns
// 9..17 fold back onto 0..8 with one multiply-add, then 2^255 = 19 at the top limb.
#pragma once

#include "coop_mont.h"
#include "edwards.h"
#include "mont.h"

namespace ell {

struct Fp25519C : FpMontC<consts::COOP_P25519, Fp25519> {
  typedef FpMontC<consts::COOP_P25519, Fp25519> B;
  typedef FpK256C R_;
  typedef B::El El;
  typedef B::W64 W64;
  typedef R_::Q Q;
  static constexpr int CL = B::CL;
  static constexpr u32 M = B::M;
  static constexpr i32 FOLD = 1216;                  // 2^261 mod p
  static constexpr bool QUAD = true;

  ELL_HD static El one() { return R_::one(); }
  // per-lane constants of the last carry pass: limb 8 keeps 23 bits (2^255 = 19 above them)
  ELL_HD static El c_sh() { return R_::each([](int l) { return 29u - (R_::m_eq(l, 8) & 6u); }); }
  ELL_HD static El c_m3() { return R_::each([](int l) { return (R_::m_lt(l, 8) & M) | (R_::m_eq(l, 8) & 0x7FFFFFu); }); }
  ELL_HD static El c_19() { return R_::each([](int l) { return R_::m_eq(l, 0) & 19u; }); }

#if defined(ELL_BOUNDS_CHECK)
  static void check(const El& a, const El& b, const char* what) {
    __int128 col[17] = {0};
    FpK256L::check_operands(R_::gather(a), R_::gather(b), col);
    const __int128 lim = ((__int128)1 << 63) - ((__int128)1 << 50);
    for (int k = 0; k < 17; k++)
      if (col[k] >= lim) { fprintf(stderr, "fp25519c %s: column %d exceeds 63 bits\n", what, k); assert(0); }
    for (int t = 9; t < CL; t++) assert(a.v[t] == 0 && b.v[t] == 0 && "fp25519c: dead lane not zero");
  }
#endif
  // carries and the fold of columns 9..17 (RW: every row folds a product of its own -- coop.h ln).
  // Bounds for column sums below 2^63: pass 1 leaves limbs below 2^29 + 2^34, pass 2 below 2^29 +
  // 2^6; column 16 splits into a 29-bit digit and a part below 2^21; the folded limbs stay below
  // 2^40, their carries below 2^11 (limb 8's, taken at bit 23: below 2^9, times 19 onto limb 0).
  // Result: limbs below 2^29 + 2^13, limb 8 below 2^23 + 2^11.
  template <bool RW = false>
  ELL_HD static El fold(const W64& acc, i64 col16) {
    const El live = B::c_live();
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
    const El cin2 = B::template up<1>(c2);
    ELL_UNROLL
    for (int t = 0; t < CL; t++) v2.v[t] = ((u32)v1.w[t] & M) + cin2.v[t];
    col16 += (i64)R_::template ln<RW, 15>(c2);
    const i32 p16 = (i32)((u32)col16 & M);
    const i32 p17 = (i32)(col16 >> 29);
    El h0 = B::template down<9>(v2);                   // lane j <- column 9 + j   (j <= 6)
    h0 = R_::template put<7>(h0, p16);
    h0 = R_::template put<8>(h0, p17);
    W64 tt;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) tt.w[t] = (i64)B::s(v2.v[t] & live.v[t]) + (i64)B::s(h0.v[t]) * FOLD;
    return carry3<RW>(tt);
  }
  // the last carry pass: nine limbs below 2^63 -> 29-bit digits + carries, limb 8 cut at bit 23 and
  // its carry folded by 19 onto limb 0
  template <bool RW = false>
  ELL_HD static El carry3(const W64& tt) {
    const El live = B::c_live(), sh = c_sh(), m3 = c_m3(), k19 = c_19();
    El c3, lo3;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) {
      c3.v[t] = (u32)(i32)(tt.w[t] >> (int)sh.v[t]);
      lo3.v[t] = (u32)tt.w[t] & m3.v[t];
    }
    const El cin3 = B::template up<1>(c3);
    const i32 hi = R_::template ln<RW, 8>(c3);
    El r;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) r.v[t] = (lo3.v[t] + cin3.v[t] + (u32)hi * k19.v[t]) & live.v[t];
    return r;
  }
  // a * k for a small constant k (a in the interface's range, 0 < k < 2^20): limbs below 2^29 + 2^25
  ELL_HD static El mul_small(const El& a, i32 k) {
    W64 tt;
    ELL_UNROLL
    for (int t = 0; t < CL; t++) tt.w[t] = (i64)B::s(a.v[t]) * (i64)k;
    return carry3<false>(tt);
  }
  ELL_HD static El mul(const El& a, const El& b) {
#if defined(ELL_BOUNDS_CHECK)
    check(a, b, "mul");
#endif
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::columns(acc, col16, a, b);
    return fold(acc, col16);
  }
  ELL_HD static El sqr(const El& a) { return mul(a, a); }
  ELL_HD static Q mulq(const Q& a, const Q& b) {
    Q r;
#if defined(__HIP_DEVICE_COMPILE__)
    W64 acc = R_::zero64();
    i64 col16 = 0;
    R_::template columns<true>(acc, col16, a.r[0], b.r[0]);
    r.r[0] = fold<true>(acc, col16);
#else
    for (int j = 0; j < R_::QR; j++) r.r[j] = mul(a.r[j], b.r[j]);
#endif
    return r;
  }
  ELL_HD static Q pack4(const El& a, const El& b, const El& c, const El& d) { return R_::pack4(a, b, c, d); }
  ELL_HD static void unpack4(const Q& q, El& a, El& b, El& c, El& d) { R_::unpack4(q, a, b, c, d); }

  // plain values: no Montgomery form here
  ELL_HD static void to_plain(u32 (&out)[8], const El& a) { B::canon(out, a); }
  ELL_HD static El from_plain(const u32 (&a)[8]) { return B::norm(R_::scatter(FpK256L::from_plain(a))); }
  // eight canonical words in memory (an entry of the one-lane kernels' tables) -> the row
  ELL_HD static El load_words(const u32* w) {
    return R_::each([&](int l) {
      const int ll = l > 8 ? 8 : l;
      const int bit = 29 * ll, k = bit >> 5, sh = bit & 31;
      const u64 two = (u64)w[k] | ((u64)(k + 1 < 8 ? w[k + 1] : 0u) << 32);
      return R_::m_lt(l, 9) & (u32)(two >> sh) & M;
    });
  }
  ELL_HD static bool is_odd(const El& a) {
    u32 w[8];
    to_plain(w, a);
    return (w[0] & 1u) != 0;
  }
  ELL_HD static bool eq(const El& a, const El& b) { return B::is_zero(B::norm(B::sub_l(a, b))); }
  static ELL_HD_NOINLINE El inv(const El& a) {
    Fp25519::El t;
    to_plain(t.v, a);
    const Fp25519::El y = Fp25519::inv(t);
    return from_plain(y.v);
  }
  template <int N>
  static ELL_HD El sqr_n(El x) {
    ELL_NOUNROLL
    for (int i = 0; i < N; i++) x = sqr(x);
    return x;
  }
  // z^(2^252 - 3)  (fp.h Fp25519::pow22523's chain)
  static ELL_HD_NOINLINE El pow22523(const El& z) {
    El z2 = sqr(z);
    El z9 = mul(sqr_n<2>(z2), z);
    El z11 = mul(z9, z2);
    El z2_5_0 = mul(sqr(z11), z9);
    El z2_10_0 = mul(sqr_n<5>(z2_5_0), z2_5_0);
    El z2_20_0 = mul(sqr_n<10>(z2_10_0), z2_10_0);
    El z2_40_0 = mul(sqr_n<20>(z2_20_0), z2_20_0);
    El z2_50_0 = mul(sqr_n<10>(z2_40_0), z2_10_0);
    El z2_100_0 = mul(sqr_n<50>(z2_50_0), z2_50_0);
    El z2_200_0 = mul(sqr_n<100>(z2_100_0), z2_100_0);
    El z2_250_0 = mul(sqr_n<50>(z2_200_0), z2_50_0);
    return mul(sqr_n<2>(z2_250_0), z);
  }
  // x with v x^2 == u, if one exists (Fp25519::sqrt_ratio's candidate and checks)
  ELL_HD static bool sqrt_ratio(El& x, const El& u, const El& v) {
    const El v3 = mul(sqr(v), v);
    const El v7 = mul(sqr(v3), v);
    const El r = mul(mul(u, v3), pow22523(mul(u, v7)));
    const El chk = mul(v, sqr(r));
    const bool ok1 = eq(chk, u);
    const bool ok2 = B::is_zero(B::norm(B::add_l(chk, u)));
    const Fp25519::El i1 = Fp25519::sqrt_m1();
    const El r2 = mul(r, from_plain(i1.v));
    x = ok1 ? r : r2;
    return ok1 || ok2;
  }
};

struct CoopEd {
  typedef Fp25519C F;
  typedef F::El El;
  typedef EdWork W1;
  typedef consts::ED25519_C C;
  static constexpr bool AVAILABLE = true;
  // (X, Y, Z, T), or a table entry's cached form (Y+X, Y-X, 2Z, 2dT)
  struct P { El a, b, c, d; };
  static constexpr int ROW_BYTES = 8 * (int)sizeof(P) * (FpK256C::CL == 1 ? 16 : 1);     // one window table of every lane
  static constexpr int ROW_BYTES2 = 2 * ROW_BYTES;                                         // two (k1*P1 + k2*P2)

  ELL_HD static P* lane_table(void* row_mem) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (P*)row_mem + (size_t)(threadIdx.x & 15u) * 16;
#else
    return (P*)row_mem;
#endif
  }
  ELL_HD static El words(const u32 (&w)[8]) {
    u32 t[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++) t[l] = w[l];
    return F::from_plain(t);
  }
  ELL_HD static P identity() {
    P r; r.a = F::zero(); r.b = F::one(); r.c = F::one(); r.d = F::zero(); return r;
  }
  ELL_HD static P from_affine(const El& x, const El& y) {
    P r; r.a = x; r.b = y; r.c = F::one(); r.d = F::mul(x, y); return r;
  }
  ELL_HD static P to_cached(const P& p) {
    P r;
    r.a = F::norm(F::add_l(p.b, p.a));
    r.b = F::norm(F::sub_l(p.b, p.a));
    r.c = F::norm(F::add_l(p.c, p.c));
    r.d = F::mul(p.d, words(C::dd));
    return r;
  }
  // -Q for a cached Q (the negated 2dT stays lazy: it is only ever a product's operand)
  ELL_HD static P cached_cneg(const P& q, bool neg) {
    P r;
    r.a = neg ? q.b : q.a;
    r.b = neg ? q.a : q.b;
    r.c = q.c;
    r.d = neg ? F::sub_l(F::zero(), q.d) : q.d;
    return r;
  }
  // _extDbl (edwards.js:174-205): two steps of four products
  ELL_HD static P dbl(const P& p) {
    const El sxy = F::norm(F::add_l(p.a, p.b));
    El A, B, Cz, S;
    F::unpack4(F::mulq(F::pack4(p.a, p.b, p.c, sxy), F::pack4(p.a, p.b, p.c, sxy)), A, B, Cz, S);
    const El E = F::norm(F::sub_l(F::sub_l(S, A), B));
    const El G = F::norm(F::sub_l(B, A));
    const El Ff = F::norm(F::sub_l(G, F::add_l(Cz, Cz)));
    const El H = F::sub_l(F::zero(), F::add_l(A, B));             // lazy: |limbs| < 2^30 + 2^14
    P r;
    F::unpack4(F::mulq(F::pack4(E, G, E, Ff), F::pack4(Ff, H, H, G)), r.a, r.b, r.d, r.c);
    return r;
  }
  // _extAdd (edwards.js:279-309), q in cached form
  ELL_HD static P add(const P& p, const P& q) {
    El A, B, Cc, D;
    F::unpack4(F::mulq(F::pack4(F::sub_l(p.b, p.a), F::add_l(p.b, p.a), p.d, p.c), F::pack4(q.b, q.a, q.d, q.c)), A, B, Cc, D);
    const El E = F::norm(F::sub_l(B, A));
    const El Ff = F::norm(F::sub_l(D, Cc));
    const El G = F::norm(F::add_l(D, Cc));
    const El H = F::add_l(B, A);                                   // lazy
    P r;
    F::unpack4(F::mulq(F::pack4(E, G, E, Ff), F::pack4(Ff, H, H, G)), r.a, r.b, r.d, r.c);
    return r;
  }
  // tbl[j-1] = cached(j*P), j = 1..8
  ELL_HD static void build_table8(P* tbl, const P& p) {
    const P pc = to_cached(p);
    tbl[0] = p;
    ELL_NOUNROLL
    for (int j = 2; j <= 8; j++) {
      P t;
      if (j & 1) t = add(tbl[j - 2], pc);
      else t = dbl(tbl[j / 2 - 1]);
      tbl[j - 1] = t;
    }
    ELL_NOUNROLL
    for (int j = 0; j < 8; j++) tbl[j] = to_cached(tbl[j]);
  }
  // signed 4-bit windows over NS tables of eight (edwards.h run_w4)
  template <int NS = 1>
  ELL_HD static P run_w4(const DigitStore& ds, const P* tbl) {
    P acc = identity();
    ELL_NOUNROLL
    for (int w = W1::NWIN - 1; w >= 0; w--) {
      if (w != W1::NWIN - 1) {
        ELL_NOUNROLL
        for (int j = 0; j < 4; j++) acc = dbl(acc);
      }
      ELL_UNROLL
      for (int s = 0; s < NS; s++) {
        const int d = ds.get(w * NS + s);
        const int ad = d < 0 ? -d : d;
        if (ad != 0) acc = add(acc, cached_cneg(tbl[s * 8 + ad - 1], d < 0));   // wave-uniform: one item per wave
      }
    }
    return acc;
  }
  // an entry of the one-lane comb table (cached form, canonical words) -> the row
  ELL_HD static P load_entry(const W1::P* e) {
    P q;
    q.a = F::load_words(e->a.v);
    q.b = F::load_words(e->b.v);
    q.c = F::load_words(e->c.v);
    q.d = F::load_words(e->d.v);
    return q;
  }
  ELL_HD static P comb_mul(const u32 (&k)[8], const W1::P* comb) {
    u32 kk[8];
    bn_copy<8>(kk, k);
    P acc = identity();
    ELL_NOUNROLL
    for (int w = 0; w < W1::COMB_W; w++) {
      const u32 d = kk[0] & (u32)W1::COMB_DIG;
      ELL_UNROLL
      for (int i = 0; i < 7; i++) kk[i] = (kk[i] >> W1::COMB_BITS) | (kk[i + 1] << (32 - W1::COMB_BITS));
      kk[7] >>= W1::COMB_BITS;
      if (d != 0) acc = add(acc, load_entry(comb + ((size_t)w * W1::COMB_DIG + (d - 1u))));
    }
    return acc;
  }
  // decodePoint (eddsa/index.js:99-109; edwards.h decode_point): x, y of the encoding; negx: -x instead
  ELL_HD static bool decode_point(P& out, const u8* enc, bool negx) {
    u32 t[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++)
      t[l] = (u32)enc[4 * l] | ((u32)enc[4 * l + 1] << 8) | ((u32)enc[4 * l + 2] << 16) | ((u32)enc[4 * l + 3] << 24);
    const bool want_odd = (t[7] >> 31) != 0;
    t[7] &= 0x7FFFFFFFu;
    const El y = F::from_plain(t);
    const El y2 = F::sqr(y);
    const El u = F::norm(F::sub_l(y2, F::one()));
    const El v = F::norm(F::add_l(F::mul(y2, words(C::d)), F::one()));
    El x;
    bool ok = F::sqrt_ratio(x, u, v);
    ok = ok && !F::is_zero(v) && !(F::is_zero(x) && want_odd);
    const bool flip = (F::is_odd(x) != want_odd) != negx;
    if (flip) x = F::norm(F::sub_l(F::zero(), x));
    out = from_affine(x, y);
    return ok;
  }
  // -> the join kernel: canonical words of X, Y, Z, limb-major, point `slot` of 2 n
  ELL_HD static void store_point(u32* ext, size_t n2, size_t slot, const P& p) {
    u32 x[8], y[8], z[8];
    F::to_plain(x, p.a);
    F::to_plain(y, p.b);
    F::to_plain(z, p.c);
    if (CoopK256::writer()) {
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        ext[(size_t)(0 * 8 + l) * n2 + slot] = x[l];
        ext[(size_t)(1 * 8 + l) * n2 + slot] = y[l];
        ext[(size_t)(2 * 8 + l) * n2 + slot] = z[l];
      }
    }
  }

  ELL_HD static P load_affine(const u8* xy, size_t i) {
    u32 tx[8], ty[8];
    load_be<8>(tx, xy + i * 64, 32);
    load_be<8>(ty, xy + i * 64 + 32, 32);
    return from_affine(F::from_plain(tx), F::from_plain(ty));
  }
  // Point#mul on G (EDDSA's a*G and r*G, KeyPair#getPublic) of one item: the comb and the item's own
  // inversion on a wave -> affine big-endian (x, y) and Point#isInfinity's flag, as edwards.h
  // normalize writes them (ed_mul_fixed + ed_normalize were two dependent one-lane launches)
  ELL_HD static void mul_fixed(size_t i, const u8* ks, const W1::P* comb, u8* out_xy, u8* out_inf) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    const P r = comb_mul(k, comb);
    // (Z = 0 cannot happen on the curve -- complete formulas; normalize puts 1 in its place)
    const El z = F::is_zero(r.c) ? F::one() : r.c;
    const El zi = F::inv(z);
    u32 x[8], y[8];
    F::to_plain(x, F::mul(r.a, zi));
    F::to_plain(y, F::mul(r.b, zi));
    bool ident = y[0] == 1u;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) ident = ident && x[l] == 0u && (l == 0 || y[l] == 0u);
    if (CoopK256::writer()) {
      store_be<8>(out_xy + i * 64, x, 32);
      store_be<8>(out_xy + i * 64 + 32, y, 32);
      if (out_inf) out_inf[i] = ident ? 1 : 0;
    }
  }
  // Point#mul (edwards.js:362-364) of one item -> ed_normalize's input (edwards.h store_ext's layout)
  ELL_HD static void mul_var(size_t i, size_t n, const u8* ks, const u8* xy, const DigitStore& ds, u32* ext,
                             void* row_mem) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    P* tbl = lane_table(row_mem);
    build_table8(tbl, load_affine(xy, i));
    recode_w4<8, W1::NNIB, true>(k, ds, 0, 1);
    store_point(ext, n, i, run_w4<1>(ds, tbl));
  }
  // Point#mulAdd (edwards.js:366-368): k1*P1 + k2*P2, or k1*G over the comb when xy1 is null
  ELL_HD static void mul_add(size_t i, size_t n, const u8* k1s, const u8* xy1, const u8* k2s, const u8* xy2,
                             const W1::P* comb, const DigitStore& ds, u32* ext, void* row_mem) {
    u32 k1[8], k2[8];
    load_be<8>(k1, k1s + i * 32, 32);
    load_be<8>(k2, k2s + i * 32, 32);
    P* tbl = lane_table(row_mem);
    if (xy1) {
      build_table8(tbl, load_affine(xy1, i));
      build_table8(tbl + 8, load_affine(xy2, i));
      recode_w4<8, W1::NNIB, true>(k1, ds, 0, 2);
      recode_w4<8, W1::NNIB, true>(k2, ds, 1, 2);
      store_point(ext, n, i, run_w4<2>(ds, tbl));
    } else {
      build_table8(tbl, load_affine(xy2, i));
      recode_w4<8, W1::NNIB, true>(k2, ds, 0, 1);
      const P b = run_w4<1>(ds, tbl);
      store_point(ext, n, i, add(comb_mul(k1, comb), to_cached(b)));
    }
  }

  // one part of EDDSA#verify (eddsa/index.js:52-63).  flags[i]: A decodes; flags[n + i]: bit 0 R
  // decodes, bit 1 S < n.
  ELL_HD static void verify_part(size_t i, size_t n, int part, const u8* msg, u64 msg_len, const u8* sig,
                                 const u8* pub, const W1::P* comb, const DigitStore& ds, u32* ext, u8* flags,
                                 void* row_mem) {
    if (part == 0) {
      u64 pre[8], st[8];
      W1::bytes_to_words(pre, sig);
      W1::bytes_to_words(pre + 4, pub);
      sha512_prefixed<8>(st, pre, msg, msg_len);
      u32 h[8];
      W1::hash_int(h, st);
      P A;
      const bool a_ok = decode_point(A, pub, false);
      P* tbl = lane_table(row_mem);
      build_table8(tbl, A);
      recode_w4<8, W1::NNIB, true>(h, ds, 0, 1);
      store_point(ext, 2 * n, i, run_w4(ds, tbl));
      if (CoopK256::writer()) flags[i] = a_ok ? 1 : 0;
    } else {
      u32 S[8], nn[8];
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        const u8* q = sig + 32 + 4 * l;
        S[l] = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
        nn[l] = C::n[l];
      }
      const bool s_ok = !bn_geq<8>(S, nn);
      P R;
      const bool r_ok = decode_point(R, sig, true);                  // -R
      const P sg = comb_mul(S, comb);
      store_point(ext, 2 * n, n + i, add(sg, to_cached(R)));
      if (CoopK256::writer()) flags[n + i] = (u8)((r_ok ? 1 : 0) | (s_ok ? 2 : 0));
    }
  }
};

// curve25519's x-only ladder for a handful of items (mont.h MontWork::ladder: mont.js:82-153), one
// item per wave.  A step is the differential addition and the doubling together: their first four
// products side by side, then three of the next four with the small-constant multiple in between,
// then x1 * (da - cb)^2.
struct CoopX25519 {
  typedef Fp25519C F;
  typedef F::El El;
  typedef MontWork W1;
  static constexpr int ROW_BYTES = 16;
  struct XZ { El x, z; };

  ELL_HD static void ladder(size_t i, size_t n, const u8* ks, const u8* xs, u32* xz) {
    u32 k[8], t[8];
    load_be<8>(k, ks + i * 32, 32);
    load_be<8>(t, xs + i * 32, 32);
    const El x1 = F::from_plain(t);
    XZ a, b;
    a.x = x1; a.z = F::one();
    b.x = F::one(); b.z = F::zero();
    ELL_NOUNROLL
    for (int w = 0; w < 256; w++) {
      const bool bit = (k[7] >> 31) != 0;                 // wave-uniform: one item per wave
      ELL_UNROLL
      for (int l = 7; l > 0; l--) k[l] = (k[l] << 1) | (k[l - 1] >> 31);
      k[0] <<= 1;
      const El as = F::norm(F::add_l(a.x, a.z)), ad = F::norm(F::sub_l(a.x, a.z));
      const El bs = F::norm(F::add_l(b.x, b.z)), bd = F::norm(F::sub_l(b.x, b.z));
      El da, cb, aa, bb;
      F::unpack4(F::mulq(F::pack4(bd, bs, bit ? as : bs, bit ? ad : bd), F::pack4(as, ad, bit ? as : bs, bit ? ad : bd)), da, cb, aa, bb);
      const El e = F::norm(F::sub_l(aa, bb));
      const El f = F::add_l(bb, F::mul_small(e, (i32)W1::C::a24[0]));      // lazy: limbs below 2^30 + 2^26
      const El sp = F::norm(F::add_l(da, cb)), sm = F::norm(F::sub_l(da, cb));
      El sx, t2, dx, dz;
      F::unpack4(F::mulq(F::pack4(sp, sm, aa, e), F::pack4(sp, sm, bb, f)), sx, t2, dx, dz);
      const El sz = F::mul(x1, t2);
      a.x = bit ? dx : sx; a.z = bit ? dz : sz;
      b.x = bit ? sx : dx; b.z = bit ? sz : dz;
    }
    u32 ox[8], oz[8];
    F::to_plain(ox, b.x);
    F::to_plain(oz, b.z);
    if (CoopK256::writer()) {
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        xz[(size_t)(0 * 8 + l) * n + i] = ox[l];
        xz[(size_t)(1 * 8 + l) * n + i] = oz[l];
      }
    }
  }
  // KeyPair#derive's pub.validate() (ec/key.js:102-105 -> mont.js:23-32) for one item on a wave of
  // its own, beside the ladder's: mont.h MontWork::x_has_point over the row field
  ELL_HD static void validate(size_t i, const u8* xs, u8* out_bad) {
    u32 t[8];
    load_be<8>(t, xs + i * 32, 32);
    const bool ok = W1::template x_has_point<F>(t);
    if (CoopK256::writer()) out_bad[i] = ok ? 0 : 1;
  }
};

}  // namespace ell
