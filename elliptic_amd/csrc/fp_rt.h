// ellgpu -- prime field with a RUN-TIME modulus (any odd p < 2^256), Montgomery form, for
// user-defined short Weierstrass curves: `new elliptic.curve.short({p, a, b, ...})` with a prime
// and coefficients of the caller's choosing (lib/elliptic/curve/short.js:11-24; bn.js gives such
// curves its generic `Mont` / `Red` contexts, dist/elliptic.js:7078-7381).  The presets' fields
// (fp.h) have their moduli as compile-time constants; here the modulus, the Montgomery
// constants and the curve's a, b come from one parameter block:
//   device  `__constant__ RtField g_rt` of the translation unit that holds the custom-curve
//           kernels (inst.hip group 16), uploaded on the call's stream before the launches;
//           scalar loads, so the limbs of p sit in SGPRs like the presets' literals do
//   host    (tests/hostsim) a plain global.
// One block per DEVICE: a call on a user-defined curve takes that device's lock, uploads the
// curve's block, and waits for its own device work before releasing the lock (Engine::CustomScope,
// engine.h) -- such calls are synchronous and serialised per device, whatever stream they name.
#pragma once

#include "fp.h"

namespace ell {

struct RtField {
  u32 p[8];        // modulus
  u32 n0;          // -p^-1 mod 2^32
  u32 one[8];      // R mod p          (R = 2^256)
  u32 r2[8];       // R^2 mod p
  u32 pm2[8];      // p - 2            (Fermat inversion)
  u32 a_m[8];      // curve coefficient a, Montgomery form
  u32 b_m[8];      // curve coefficient b, Montgomery form
  u32 a_kind;      // 0: a == 0, 3: a == p - 3, 1: anything else
  u32 d_m[8];      // Edwards curves (edcustom.h): coefficient d, Montgomery form; a_m holds a
  u32 kind;        // 0: short Weierstrass (a_m, b_m), 1: (twisted) Edwards with c = 1 (a_m, d_m)
};

#if defined(__HIP_DEVICE_COMPILE__)
extern __constant__ RtField g_rt;
#define ELL_RT (g_rt)
#else
inline RtField& rt_host_block() { static RtField f; return f; }
#define ELL_RT (rt_host_block())
#endif

struct FpMontRT {
  static constexpr int L = 8;
  typedef Fe<8> El;
  static constexpr bool HAS_SQRT = false;      // Red#sqrt of a generic prime stays in the reference's JavaScript

  ELL_HD static void get_p(u32 (&p)[8]) {
    ELL_UNROLL
    for (int i = 0; i < 8; i++) p[i] = ELL_RT.p[i];
  }
  ELL_HD static El zero() { El r; bn_zero<8>(r.v); return r; }
  ELL_HD static El one() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = ELL_RT.one[i];
    return r;
  }
  ELL_HD static El curve_a() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = ELL_RT.a_m[i];
    return r;
  }
  ELL_HD static El curve_b() {
    El r;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r.v[i] = ELL_RT.b_m[i];
    return r;
  }
  ELL_HD static bool is_zero(const El& a) { return bn_is_zero<8>(a.v); }
  ELL_HD static bool eq(const El& a, const El& b) { return bn_eq<8>(a.v, b.v); }
  ELL_HD static El add(const El& a, const El& b) {
    u32 p[8]; get_p(p);
    El r; mod_add<8>(r.v, a.v, b.v, p); return r;
  }
  ELL_HD static El sub(const El& a, const El& b) {
    u32 p[8]; get_p(p);
    El r; mod_sub<8>(r.v, a.v, b.v, p); return r;
  }
  ELL_HD static El neg(const El& a) { return sub(zero(), a); }
  ELL_HD static El dbl(const El& a) { return add(a, a); }
  template <int K>
  ELL_HD static El mul_pow2(const El& a) {
    El r = dbl(a);
    ELL_UNROLL
    for (int i = 1; i < K; i++) r = dbl(r);
    return r;
  }
  // Montgomery reduction of a 16-limb value (row-wise, explicit carry chains; FpMont::redc with the
  // modulus read from the parameter block)
  ELL_HD static El redc(u32 (&t)[16]) {
    u32 p[8]; get_p(p);
    const u32 n0 = ELL_RT.n0;
    u32 top = 0;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) {
      u32 m = t[i] * n0;
      u32 lo[8], hi[8];
      ELL_UNROLL
      for (int j = 0; j < 8; j++) {
        u64 x = (u64)m * p[j];
        lo[j] = (u32)x;
        hi[j] = (u32)(x >> 32);
      }
      u32 u[9];
      u32 c = 0;
      u[0] = lo[0];
      ELL_UNROLL
      for (int j = 1; j < 8; j++) u[j] = addc32(lo[j], hi[j - 1], c, c);
      u[8] = hi[7] + c;
      c = 0;
      ELL_UNROLL
      for (int j = 0; j < 8; j++) t[i + j] = addc32(t[i + j], u[j], c, c);
      u32 c1, c2;
      u32 y = addc32(t[i + 8], u[8], c, c1);
      t[i + 8] = addc32(y, 0, top, c2);
      top = c1 + c2;
    }
    u32 r[8], sres[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r[i] = t[8 + i];
    u32 br = bn_sub<8>(sres, r, p);
    El out;
    bn_select<8>(out.v, (top != 0) || (br == 0), sres, r);
    return out;
  }
  ELL_HD static El mul(const El& a, const El& b) {
    u32 t[16];
    fe_mul_wide<8>(t, a.v, b.v);
    return redc(t);
  }
  ELL_HD static El sqr(const El& a) {
    u32 t[16];
    fe_sqr_wide<8>(t, a.v);
    return redc(t);
  }
  ELL_HD static El sqr_n(El a, int n) {
    ELL_NOUNROLL
    for (int i = 0; i < n; i++) a = sqr(a);
    return a;
  }
  ELL_HD static El from_plain(const u32 (&a)[8]) {       // a < 2^256 (possibly >= p): a * R mod p
    El x, r2;
    bn_copy<8>(x.v, a);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) r2.v[i] = ELL_RT.r2[i];
    return mul(x, r2);
  }
  ELL_HD static void to_plain(u32 (&r)[8], const El& a) {
    El o; bn_zero<8>(o.v); o.v[0] = 1;
    El x = mul(a, o);
    bn_copy<8>(r, x.v);
  }
  ELL_HD static bool is_odd(const El& a) { u32 r[8]; to_plain(r, a); return r[0] & 1; }
  // a^(p-2): a^-1, 0 for 0 (the exponent is wave-uniform: a scalar branch per bit)
  static ELL_HD_NOINLINE El inv(const El& a) {
    El r = one();
    ELL_NOUNROLL
    for (int i = 255; i >= 0; i--) {
      r = sqr(r);
      if ((ELL_RT.pm2[i >> 5] >> (i & 31)) & 1u) r = mul(r, a);
    }
    return r;
  }
  static ELL_HD_NOINLINE El sqrt(const El& a) { return a; }     // never called (HAS_SQRT = false)
};

}  // namespace ell
