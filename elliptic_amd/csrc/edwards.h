// ellgpu -- ed25519 (twisted Edwards, a = -1) in extended coordinates.
//
// Replaces lib/elliptic/curve/edwards.js Point#mul / mulAdd / jmulAdd
// (:362-375) and the base.js ladders they call, plus normalize (:377-390):
//   dbl       <- _extDbl (edwards.js:174-205, dbl-2008-hwcd, 4M+4S)
//   add       <- _extAdd (edwards.js:279-309, add-2008-hwcd-3, 8M) with the
//                second operand pre-transformed to (Y+X, Y-X, 2Z, 2dT)
// The a = -1 addition law is complete on ed25519 (d is a non-square), so a
// lane never needs an exceptional branch: identity, doubling and inverse pairs
// all come out of the same formula.  Ladder shape as in ladder.h: signed 4-bit
// fixed windows for a variable base, 8-bit comb for G.
#pragma once

#include "curve_consts.h"
#include "ladder.h"
#include "sha512.h"
#include "hmac_drbg512.h"

#ifndef ELL_COMB_BITS_256
#define ELL_COMB_BITS_256 16
#endif

namespace ell {

struct EdWork {
  typedef Fp25519 F;
  typedef F::El El;
  typedef consts::ED25519_C C;
  static constexpr int L = 8;
  static constexpr int BYTES = 32;
  static constexpr int NNIB = 64;
  static constexpr int NWIN = 65;              // 64 signed windows + the carry window
  static constexpr int COMB_BITS = ELL_COMB_BITS_256;   // 16-bit comb windows: 16 adds per k*G
  static constexpr int COMB_W = 256 / COMB_BITS;
  static constexpr int COMB_DIG = (1 << COMB_BITS) - 1;
  static constexpr size_t COMB_ENTRIES = (size_t)COMB_W * COMB_DIG;

  // Extended point (X, Y, Z, T) -- or, for table entries, the "cached" form
  // (Y+X, Y-X, 2Z, 2dT) stored in the same four slots.
  struct P {
    El a, b, c, d;
  };

  ELL_HD static El const_dd() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = C::dd[i];
    return r;
  }
  ELL_HD static P identity() {
    P r; r.a = F::zero(); r.b = F::one(); r.c = F::one(); r.d = F::zero(); return r;
  }
  ELL_HD static P from_affine(const El& x, const El& y) {
    P r; r.a = x; r.b = y; r.c = F::one(); r.d = F::mul(x, y); return r;
  }
  ELL_HD static P to_cached(const P& p) {
    P r;
    r.a = F::add(p.b, p.a);
    r.b = F::sub(p.b, p.a);
    r.c = F::template mul_pow2<1>(p.c);
    r.d = F::mul(p.d, const_dd());
    return r;
  }
  // -Q for a cached Q: swap (Y+X, Y-X), negate 2dT
  ELL_HD static P cached_cneg(const P& q, bool neg) {
    P r;
    r.a = fe_select<F>(neg, q.b, q.a);
    r.b = fe_select<F>(neg, q.a, q.b);
    r.c = q.c;
    r.d = fe_select<F>(neg, F::neg(q.d), q.d);
    return r;
  }
  ELL_HD static P select(bool c, const P& x, const P& y) {
    P r;
    r.a = fe_select<F>(c, x.a, y.a);
    r.b = fe_select<F>(c, x.b, y.b);
    r.c = fe_select<F>(c, x.c, y.c);
    r.d = fe_select<F>(c, x.d, y.d);
    return r;
  }
  // WT = false: the T coordinate of the result is not produced (only an addition reads it, so
  // a doubling followed by a doubling leaves it out: 3M + 4S)
  template <bool WT = true>
  ELL_HD static P dbl(const P& p) {
    El A = F::sqr(p.a);
    El B = F::sqr(p.b);
    El Cc = F::template mul_pow2<1>(F::sqr(p.c));
    El D = F::neg(A);
    El E = F::sub(F::sub(F::sqr(F::add(p.a, p.b)), A), B);
    El G = F::add(D, B);
    El Ff = F::sub(G, Cc);
    El H = F::sub(D, B);
    P r;
    r.a = F::mul(E, Ff);
    r.b = F::mul(G, H);
    if (WT) r.d = F::mul(E, H);
    else r.d = p.d;
    r.c = F::mul(Ff, G);
    return r;
  }
  // p (extended) + q (cached); !do_add returns p.  with_t = false: T of the sum is not produced
  // (the next operation is a doubling)
  ELL_HD static P add(const P& p, const P& q, bool do_add = true, bool with_t = true) {
    El A = F::mul(F::sub(p.b, p.a), q.b);
    El B = F::mul(F::add(p.b, p.a), q.a);
    El Cc = F::mul(p.d, q.d);
    El D = F::mul(p.c, q.c);
    El E = F::sub(B, A);
    El Ff = F::sub(D, Cc);
    El G = F::add(D, Cc);
    El H = F::add(B, A);
    P r;
    r.a = F::mul(E, Ff);
    r.b = F::mul(G, H);
    r.d = p.d;
    if (with_t) r.d = F::mul(E, H);               // wave-uniform condition
    r.c = F::mul(Ff, G);
    return select(do_add, r, p);
  }

  ELL_HD static El load_fe(const u8* p) {
    u32 t[8];
    load_be<8>(t, p, 32);
    return F::from_plain(t);
  }
  ELL_HD static P load_affine(const u8* xy, size_t i) {
    El x = load_fe(xy + i * 64);
    El y = load_fe(xy + i * 64 + 32);
    return from_affine(x, y);
  }
  ELL_HD static void store_ext(u32* ext, size_t n, size_t i, const P& p) {
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {
      ext[(size_t)(0 * 8 + l) * n + i] = p.a.v[l];
      ext[(size_t)(1 * 8 + l) * n + i] = p.b.v[l];
      ext[(size_t)(2 * 8 + l) * n + i] = p.c.v[l];
    }
  }

  // tbl[j-1] = cached(j*P), j = 1..8
  ELL_HD static void build_table8(P* tbl, const P& p) {
    P pc = to_cached(p);
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

  template <int NS>
  ELL_HD static P run_w4(const DigitStore& ds, const P* tbl) {
    P acc = identity();
    ELL_NOUNROLL
    for (int w = NWIN - 1; w >= 0; w--) {
      if (w != NWIN - 1) {
        ELL_NOUNROLL
        for (int j = 0; j < 3; j++) acc = dbl<false>(acc);
        acc = dbl<true>(acc);                      // the addition below reads T
      }
      ELL_UNROLL
      for (int s = 0; s < NS; s++) {
        int d = ds.get(w * NS + s);
        int ad = d < 0 ? -d : d;
        int e = ad ? ad - 1 : 0;
        P q = cached_cneg(tbl[s * 8 + e], d < 0);
        // T of the sum is read by the next addition of this window, or by the caller after the
        // last window; four doublings follow otherwise
        acc = add(acc, q, ad != 0, s + 1 < NS || w == 0);
      }
    }
    return acc;
  }

  ELL_HD static P comb_mul(const u32 (&k)[8], const P* comb) {
    u32 kk[8];
    bn_copy<8>(kk, k);
    P acc = identity();
    ELL_NOUNROLL
    for (int w = 0; w < COMB_W; w++) {
      u32 d = kk[0] & (u32)COMB_DIG;
      ELL_UNROLL
      for (int i = 0; i < 7; i++) kk[i] = (kk[i] >> COMB_BITS) | (kk[i + 1] << (32 - COMB_BITS));
      kk[7] >>= COMB_BITS;
      u32 e = d ? d - 1 : 0;
      acc = add(acc, comb[(size_t)w * COMB_DIG + e], d != 0);
    }
    return acc;
  }

  // ---- work items ----------------------------------------------------------------
  ELL_HD static void mul_var(size_t i, size_t n, const u8* ks, const u8* xy, P* tbl_all,
                             const DigitStore& ds, u32* ext) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    P p = load_affine(xy, i);
    P* tbl = tbl_all + i * 8;
    build_table8(tbl, p);
    recode_w4<8, NNIB, true>(k, ds, 0, 1);
    store_ext(ext, n, i, run_w4<1>(ds, tbl));
  }
  ELL_HD static void mul_fixed(size_t i, size_t n, const u8* ks, const P* comb, u32* ext) {
    u32 k[8];
    load_be<8>(k, ks + i * 32, 32);
    store_ext(ext, n, i, comb_mul(k, comb));
  }
  ELL_HD static void mul_add_g(size_t i, size_t n, const u8* k1s, const u8* k2s, const u8* xy2,
                               const P* comb, P* tbl_all, const DigitStore& ds, u32* ext) {
    u32 k1[8], k2[8];
    load_be<8>(k1, k1s + i * 32, 32);
    load_be<8>(k2, k2s + i * 32, 32);
    P p2 = load_affine(xy2, i);
    P* tbl = tbl_all + i * 8;
    build_table8(tbl, p2);
    recode_w4<8, NNIB, true>(k2, ds, 0, 1);
    P b = run_w4<1>(ds, tbl);
    P a = comb_mul(k1, comb);
    store_ext(ext, n, i, add(a, to_cached(b)));
  }
  ELL_HD static void mul_add2(size_t i, size_t n, const u8* k1s, const u8* xy1, const u8* k2s,
                              const u8* xy2, P* tbl_all, const DigitStore& ds, u32* ext) {
    u32 k1[8], k2[8];
    load_be<8>(k1, k1s + i * 32, 32);
    load_be<8>(k2, k2s + i * 32, 32);
    P* tbl = tbl_all + i * 16;
    build_table8(tbl, load_affine(xy1, i));
    build_table8(tbl + 8, load_affine(xy2, i));
    recode_w4<8, NNIB, true>(k1, ds, 0, 2);
    recode_w4<8, NNIB, true>(k2, ds, 1, 2);
    store_ext(ext, n, i, run_w4<2>(ds, tbl));
  }

  // EdwardsCurve#pointFromY (edwards.js:71-97, c = 1, a = -1): x^2 = (y^2 - 1)/(d y^2 + 1),
  // x with the requested parity.  ok = 0 when there is no such point -- also for the
  // reference's two throwing corner cases: x = 0 with odd requested, and d y^2 + 1 == 0.
  ELL_HD static void decompress(size_t i, const u8* ys, const u8* odd, u8* out_xy, u8* out_ok) {
    if (odd[i] & 2u) { from_x(i, ys, odd, out_xy, out_ok); return; }
    El y = load_fe(ys + i * 32);
    El d;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) d.v[l] = C::d[l];
    El y2 = F::sqr(y);
    El u = F::sub(y2, F::one());
    El v = F::add(F::mul(y2, d), F::one());
    El x;
    bool ok = F::sqrt_ratio(x, u, v);
    bool want_odd = (odd[i] & 1u) != 0;
    bool xzero = F::is_zero(x);
    ok = ok && !F::is_zero(v) && !(xzero && want_odd);
    bool is_odd = (x.v[0] & 1u) != 0;
    El xn = F::neg(x);
    bn_select<8>(x.v, is_odd != want_odd, xn.v, x.v);
    if (!ok) { x = F::zero(); y = F::zero(); }
    store_be<8>(out_xy + i * 64, x.v, 32);
    store_be<8>(out_xy + i * 64 + 32, y.v, 32);
    out_ok[i] = ok ? 1 : 0;
  }
  // EdwardsCurve#pointFromX (edwards.js:50-69, c = 1, a = -1): y^2 = (1 + x^2)/(1 - d x^2), y with
  // the requested parity (no special case for y = 0: -0 = 0 is what redNeg gives).  1 - d x^2
  // is never 0 (d is not a square).  ok = 0 where the reference throws 'invalid point'.
  ELL_HD static void from_x(size_t i, const u8* xs, const u8* odd, u8* out_xy, u8* out_ok) {
    El x = load_fe(xs + i * 32);
    El d;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) d.v[l] = C::d[l];
    El x2 = F::sqr(x);
    El u = F::add(F::one(), x2);
    El v = F::sub(F::one(), F::mul(x2, d));
    El y;
    bool ok = F::sqrt_ratio(y, u, v);
    bool want_odd = (odd[i] & 1u) != 0;
    bool is_odd = (y.v[0] & 1u) != 0;
    El yn = F::neg(y);
    bn_select<8>(y.v, is_odd != want_odd, yn.v, y.v);
    if (!ok) { x = F::zero(); y = F::zero(); }
    store_be<8>(out_xy + i * 64, x.v, 32);
    store_be<8>(out_xy + i * 64 + 32, y.v, 32);
    out_ok[i] = ok ? 1 : 0;
  }

  // decodePoint (eddsa/index.js:99-109) of a 32-byte little-endian encoding: y with the
  // top bit cleared (reduced mod p like BN#toRed), the top bit is x's parity.
  ELL_HD static bool decode_point(P& out, const u8* enc) {
    u32 t[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++)
      t[l] = (u32)enc[4 * l] | ((u32)enc[4 * l + 1] << 8) | ((u32)enc[4 * l + 2] << 16) |
             ((u32)enc[4 * l + 3] << 24);
    bool want_odd = (t[7] >> 31) != 0;
    t[7] &= 0x7FFFFFFFu;
    El y = F::from_plain(t);
    El d;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) d.v[l] = C::d[l];
    El y2 = F::sqr(y);
    El u = F::sub(y2, F::one());
    El v = F::add(F::mul(y2, d), F::one());
    El x;
    bool ok = F::sqrt_ratio(x, u, v);
    ok = ok && !F::is_zero(v) && !(F::is_zero(x) && want_odd);
    bool is_odd = (x.v[0] & 1u) != 0;
    El xn = F::neg(x);
    bn_select<8>(x.v, is_odd != want_odd, xn.v, x.v);
    out = from_affine(x, y);
    return ok;
  }

  // Point#add (edwards.js:350-360 -> _extAdd :279-309) for affine inputs -> extended; a set
  // inf flag stands for the identity (0, 1)
  ELL_HD static void point_add(size_t i, size_t n, const u8* xy1, const u8* inf1, const u8* xy2,
                               const u8* inf2, u32* ext) {
    P id = from_affine(F::zero(), F::one());
    P p = load_affine(xy1, i);
    P q = load_affine(xy2, i);
    if (inf1 && inf1[i]) p = id;
    if (inf2 && inf2[i]) q = id;
    store_ext(ext, n, i, add(p, to_cached(q)));
  }

  // ---- batch codecs / validation ---------------------------------------------------
  // EDDSA#decodePoint (eddsa/index.js:99-109): status 0 = point, 2 = 'invalid point'
  ELL_HD static void decode_points(size_t i, const u8* enc, u8* out_xy, u8* status) {
    P pt;
    bool ok = decode_point(pt, enc + i * 32);
    if (!ok) { pt.a = F::zero(); pt.b = F::zero(); }
    store_be<8>(out_xy + i * 64, pt.a.v, 32);
    store_be<8>(out_xy + i * 64 + 32, pt.b.v, 32);
    status[i] = ok ? 0 : 2;
  }
  // EDDSA#encodePoint (eddsa/index.js:94-98) of affine x || y (big-endian, reduced mod p)
  ELL_HD static void encode_points(size_t i, const u8* xy, u8* out) {
    El x = load_fe(xy + i * 64), y = load_fe(xy + i * 64 + 32);
    u8 can[64];
    store_be<8>(can, x.v, 32);
    store_be<8>(can + 32, y.v, 32);
    u8 enc[32];
    encode_affine(enc, can);
    ELL_UNROLL
    for (int j = 0; j < 32; j++) out[i * 32 + j] = enc[j];
  }
  // EdwardsCurve#validate (edwards.js:99-112) of an affine point: a x^2 + y^2 == 1 + d x^2 y^2
  // (a = -1, c = 1).  status 0 = on the curve, 2 = not a point, 1 = flagged as the identity by
  // the caller (`inf`, for symmetry with the short curves' KeyPair#validate).
  ELL_HD static bool on_curve(const El& x, const El& y) {
    El d;
    ELL_UNROLL
    for (int l = 0; l < 8; l++) d.v[l] = C::d[l];
    El x2 = F::sqr(x), y2 = F::sqr(y);
    El lhs = F::sub(y2, x2);
    El rhs = F::add(F::one(), F::mul(F::mul(d, x2), y2));
    return F::eq(lhs, rhs);
  }
  ELL_HD static void validate_point(size_t i, const u8* xy, const u8* inf, u8* status) {
    if (inf && inf[i]) { status[i] = 1; return; }
    El x = load_fe(xy + i * 64), y = load_fe(xy + i * 64 + 32);
    status[i] = on_curve(x, y) ? 0 : 2;
  }
  // The engine's domain is points ON the curve (work.h: Work::on_curve has the reasoning; here the
  // reference's doubling, edwards.js:183-205, even differs from its addition of a point to itself
  // off the curve).  After the ladder + normalization of a point-valued call: items with an
  // operand (xy1 / xy2, either may be null) that is not on the curve get out_inf = 2 and a zeroed
  // result, never a guessed one.
  ELL_HD static void domain_mark(size_t i, const u8* xy1, const u8* xy2, u8* out_xy, u8* out_inf) {
    bool on = true;
    if (xy1) on = on_curve(load_fe(xy1 + i * 64), load_fe(xy1 + i * 64 + 32));
    if (xy2) on = on && on_curve(load_fe(xy2 + i * 64), load_fe(xy2 + i * 64 + 32));
    if (on) return;
    out_inf[i] = 2;
    if (out_xy) {
      ELL_NOUNROLL
      for (int b = 0; b < 64; b++) out_xy[i * 64 + b] = 0;
    }
  }
  ELL_HD static void fill_order(size_t i, u8* scal) {
    typedef FpMont<consts::ED25519_N> Fn;
    u32 nn[8];
    Fn::get_p(nn);
    store_be<8>(scal + i * 32, nn, 32);
  }

  // EDDSA#verify (eddsa/index.js:52-63): S < n, h = SHA-512(R || A || M) mod n (hashInt
  // :65-70, little-endian), accept iff R + h*A == S*G.  ok = 0/1; err = 1 where the
  // reference throws (R or A does not decode to a curve point) -- only evaluated when S < n,
  // as in the reference.  sig = R || S and pub = A in their 32-byte wire encodings.
  // hashInt (eddsa/index.js:65-70): a 64-byte digest read little-endian, mod n -> plain limbs.
  // h = (hi * 2^256 + lo) mod n: the Montgomery image of hi IS hi * 2^256 mod n.
  // The digest arrives as SHA-512's eight big-endian 64-bit state words; little-endian limb 2k
  // of the byte string is the byte-swapped high half of word k, limb 2k + 1 the low half.
  ELL_HD static void digest_limbs(u32 (&lo)[8], u32 (&hi)[8], const u64 (&st)[8]) {
    ELL_UNROLL
    for (int k = 0; k < 4; k++) {
      lo[2 * k] = __builtin_bswap32((u32)(st[k] >> 32));
      lo[2 * k + 1] = __builtin_bswap32((u32)st[k]);
      hi[2 * k] = __builtin_bswap32((u32)(st[4 + k] >> 32));
      hi[2 * k + 1] = __builtin_bswap32((u32)st[4 + k]);
    }
  }
  ELL_HD static void hash_int(u32 (&h)[8], const u64 (&st)[8]) {
    typedef FpMont<consts::ED25519_N> Fn;
    u32 lo[8], hi[8];
    digest_limbs(lo, hi, st);
    Fn::El hm = Fn::from_plain(hi);
    u32 lor[8];
    Fn::to_plain(lor, Fn::from_plain(lo));
    Fn::El lom;
    bn_copy<8>(lom.v, lor);
    Fn::El hsum = Fn::add(hm, lom);
    bn_copy<8>(h, hsum.v);
  }
  // the four big-endian 64-bit words of a 32-byte string given as little-endian limbs
  ELL_HD static void le_limbs_to_words(u64* w, const u32 (&t)[8]) {
    ELL_UNROLL
    for (int k = 0; k < 4; k++)
      w[k] = ((u64)__builtin_bswap32(t[2 * k]) << 32) | (u64)__builtin_bswap32(t[2 * k + 1]);
  }
  // ... and of 32 bytes in memory
  ELL_HD static void bytes_to_words(u64* w, const u8* p) {
    ELL_UNROLL
    for (int k = 0; k < 4; k++) {
      u64 x;
      __builtin_memcpy(&x, p + 8 * k, 8);
      w[k] = __builtin_bswap64(x);
    }
  }

  // ---- EdDSA sign (eddsa/index.js:32-50 with KeyPair.fromSecret, eddsa/key.js:42-75) ----
  // stage 1: hash = SHA-512(secret); a = the clamped first half (key.js:51-62), prefix = the
  // second half; r = hashInt(prefix || M).  Both scalars go out as 32-byte big-endian inputs
  // of the fixed-base kernel: scal[i] = a, scal[n + i] = r.
  ELL_HD static void sign_pre(size_t i, size_t n, const u8* secret, const u8* msg, u64 msg_len, u8* scal) {
    u64 hs[8];
    sha512_prefixed<0>(hs, nullptr, secret, 32);
    u32 a[8], pf[8];
    digest_limbs(a, pf, hs);
    a[0] &= ~7u;                                   // hash[0] &= 248
    a[7] = (a[7] & 0x7FFFFFFFu) | 0x40000000u;     // hash[31] &= 127, |= 64
    store_be<8>(scal + i * 32, a, 32);
    u64 st[8];
    sha512_prefixed<4>(st, hs + 4, msg, msg_len);
    u32 r[8];
    hash_int(r, st);
    store_be<8>(scal + (n + i) * 32, r, 32);
  }
  // encodePoint (eddsa/index.js:94-98) of an affine point given as x || y big-endian
  ELL_HD static void encode_affine(u8 (&enc)[32], const u8* xy) {
    ELL_UNROLL
    for (int j = 0; j < 32; j++) enc[j] = xy[32 + 31 - j];
    enc[31] |= (u8)((xy[31] & 1) << 7);
  }
  // stage 3: A = a G and R = r G arrive as affine points; h = hashInt(Renc || Aenc || M),
  // S = (r + h a) mod n; sig = Renc || S (little-endian), pub = Aenc
  ELL_HD static void sign_post(size_t i, size_t n, const u8* msg, u64 msg_len, const u8* scal,
                               const u8* xy, u8* sig, u8* pub) {
    typedef FpMont<consts::ED25519_N> Fn;
    // encodePoint: y little-endian with x's parity in the top bit, as limbs
    u32 ae[8], re[8], xa[8], xr[8];
    load_be<8>(xa, xy + i * 64, 32);
    load_be<8>(ae, xy + i * 64 + 32, 32);
    load_be<8>(xr, xy + (n + i) * 64, 32);
    load_be<8>(re, xy + (n + i) * 64 + 32, 32);
    ae[7] |= (xa[0] & 1u) << 31;
    re[7] |= (xr[0] & 1u) << 31;
    u64 pre[8];
    le_limbs_to_words(pre, re);
    le_limbs_to_words(pre + 4, ae);
    u64 st[8];
    sha512_prefixed<8>(st, pre, msg, msg_len);
    u32 h[8], a[8], r[8];
    hash_int(h, st);
    load_be<8>(a, scal + i * 32, 32);
    load_be<8>(r, scal + (n + i) * 32, 32);
    u32 ha[8];
    Fn::to_plain(ha, Fn::mul(Fn::from_plain(h), Fn::from_plain(a)));
    Fn::El x, y;
    bn_copy<8>(x.v, ha);
    bn_copy<8>(y.v, r);
    Fn::El S = Fn::add(x, y);                     // both canonical residues: plain modular add
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {                  // little-endian limbs = the wire bytes
      __builtin_memcpy(sig + i * 64 + 4 * l, &re[l], 4);
      __builtin_memcpy(sig + i * 64 + 32 + 4 * l, &S.v[l], 4);
    }
    if (pub) {
      ELL_UNROLL
      for (int l = 0; l < 8; l++) __builtin_memcpy(pub + i * 32 + 4 * l, &ae[l], 4);
    }
  }

  ELL_HD static void eddsa_verify(size_t i, const u8* msg, u64 msg_len, const u8* sig,
                                  const u8* pub, const P* comb, P* tbl_all, const DigitStore& ds,
                                  u8* out_ok, u8* out_err) {
    typedef FpMont<consts::ED25519_N> Fn;
    u32 S[8], nn[8];
    ELL_UNROLL
    for (int l = 0; l < 8; l++) {
      const u8* q = sig + 32 + 4 * l;
      S[l] = (u32)q[0] | ((u32)q[1] << 8) | ((u32)q[2] << 16) | ((u32)q[3] << 24);
      nn[l] = C::n[l];
    }
    bool s_ok = !bn_geq<8>(S, nn);
    u64 pre[8], st[8];
    bytes_to_words(pre, sig);
    bytes_to_words(pre + 4, pub);
    sha512_prefixed<8>(st, pre, msg, msg_len);
    u32 h[8];
    hash_int(h, st);

    P A, R;
    bool a_ok = decode_point(A, pub);
    bool r_ok = decode_point(R, sig);
    P* tbl = tbl_all + i * 8;
    build_table8(tbl, A);
    recode_w4<8, NNIB, true>(h, ds, 0, 1);
    P hA = run_w4<1>(ds, tbl);
    P lhs = add(hA, to_cached(R));
    P SG = comb_mul(S, comb);
    // projective equality (Point#eq compares the normalized coordinates, edwards.js:409-413)
    bool same = F::eq(F::mul(lhs.a, SG.c), F::mul(SG.a, lhs.c)) &&
                F::eq(F::mul(lhs.b, SG.c), F::mul(SG.b, lhs.c));
    bool bad = s_ok && !(a_ok && r_ok);
    out_ok[i] = (s_ok && a_ok && r_ok && same) ? 1 : 0;
    if (out_err) out_err[i] = bad ? 1 : 0;
  }

  // the comparison of EDDSA#verify for the items whose two sides came from the lanes-per-item layer
  // (coop_ed.h verify_part): ext holds h*A (point i) and S*G - R (point n + i) as canonical X, Y, Z
  // words; flags as verify_part writes them.  Point#eq (edwards.js:409-413), projectively.
  ELL_HD static void eddsa_join(size_t i, size_t n, const u32* ext, const u8* flags, u8* out_ok, u8* out_err) {
    El c[6];
    ELL_UNROLL
    for (int k = 0; k < 6; k++) {
      const size_t slot = (k < 3 ? 0 : n) + i;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) c[k].v[l] = ext[(size_t)((k % 3) * 8 + l) * (2 * n) + slot];
    }
    const bool same = F::eq(F::mul(c[0], c[5]), F::mul(c[3], c[2])) && F::eq(F::mul(c[1], c[5]), F::mul(c[4], c[2]));
    const bool a_ok = (flags[i] & 1u) != 0, r_ok = (flags[n + i] & 1u) != 0, s_ok = (flags[n + i] & 2u) != 0;
    out_ok[i] = (s_ok && a_ok && r_ok && same) ? 1 : 0;
    if (out_err) out_err[i] = (s_ok && !(a_ok && r_ok)) ? 1 : 0;
  }

  // (X:Y:Z) -> affine (x, y) with one inversion per K items (normalize,
  // edwards.js:377-390).  out_inf mirrors Point#isInfinity (edwards.js:167-172):
  // x == 0 && y == 1.  raw != null stores cached(x, y, 1, xy) for the comb.
  ELL_HD static void normalize(size_t t, size_t T, size_t n, int K, const u32* ext, u32* pre,
                               u8* out_xy, u8* out_inf, P* raw) {
    El acc = F::one();
    ELL_NOUNROLL
    for (int j = 0; j < K; j++) {
      size_t i = t + (size_t)j * T;
      if (i >= n) break;
      El z;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) z.v[l] = ext[(size_t)(2 * 8 + l) * n + i];
      // Z == 0 cannot happen for a point on the curve (complete formulas); an
      // off-curve input must still not poison the other K-1 items of the batch
      z = fe_select<F>(F::is_zero(z), F::one(), z);
      ELL_UNROLL
      for (int l = 0; l < 8; l++) pre[(size_t)l * n + i] = acc.v[l];
      acc = F::mul(acc, z);
    }
    El inv = F::inv(acc);
    ELL_NOUNROLL
    for (int j = K - 1; j >= 0; j--) {
      size_t i = t + (size_t)j * T;
      if (i >= n) continue;
      El X, Y, Z, pr;
      ELL_UNROLL
      for (int l = 0; l < 8; l++) {
        X.v[l] = ext[(size_t)(0 * 8 + l) * n + i];
        Y.v[l] = ext[(size_t)(1 * 8 + l) * n + i];
        Z.v[l] = ext[(size_t)(2 * 8 + l) * n + i];
        pr.v[l] = pre[(size_t)l * n + i];
      }
      Z = fe_select<F>(F::is_zero(Z), F::one(), Z);
      El zinv = F::mul(inv, pr);
      inv = F::mul(inv, Z);
      El x = F::mul(X, zinv);
      El y = F::mul(Y, zinv);
      if (out_xy) {
        store_be<8>(out_xy + i * 64, x.v, 32);
        store_be<8>(out_xy + i * 64 + 32, y.v, 32);
      }
      if (out_inf) out_inf[i] = (F::is_zero(x) && F::eq(y, F::one())) ? 1 : 0;
      if (raw) raw[i] = to_cached(from_affine(x, y));
    }
  }
};

}  // namespace ell
