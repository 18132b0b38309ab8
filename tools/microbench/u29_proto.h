// prototype: secp256k1 field, 9 limbs x 29 bits, signed limbs
#pragma once
#include <stdint.h>
typedef uint32_t u32; typedef int32_t i32; typedef uint64_t u64; typedef int64_t i64;
#ifdef __HIPCC__
#define HD __host__ __device__ __forceinline__
#else
#define HD inline
#endif
// acc + a*b as ONE v_mad_i64_i32 whose addend is acc: written as asm because the compiler, left to
// itself, starts every column's chain at zero (a shorter dependency chain) and adds the incoming
// carry with a separate 64-bit add -- one more VOP3 instruction per column
#ifdef __HIP_DEVICE_COMPILE__
HD i64 smad(i64 acc, i32 a, i32 b) {
  i64 r;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=&v"(r) : "v"(a), "v"(b), "v"(acc) : "vcc");
  return r;
}
HD i64 smadk(i64 acc, i32 a, i32 k) {          // k in an SGPR / inline constant
  i64 r;
  asm("v_mad_i64_i32 %0, vcc, %1, %2, %3" : "=&v"(r) : "v"(a), "s"(k), "v"(acc) : "vcc");
  return r;
}
#else
HD i64 smad(i64 acc, i32 a, i32 b) { return acc + (i64)a * b; }
HD i64 smadk(i64 acc, i32 a, i32 k) { return acc + (i64)a * k; }
#endif
struct U29 {
  static constexpr u32 M = (1u << 29) - 1;
  static constexpr i32 R0 = 31264;                // 2^261 = R1 * 2^29 + R0 (mod p)
#ifdef __HIP_DEVICE_COMPILE__
  // R1 = 256 read through an opaque move: as a literal the compiler turns h * 256 into a 64-bit
  // shift + a 64-bit add (two VOP3 instructions) instead of one multiply-add
  HD static i32 r1() { return 256; }
#else
  HD static i32 r1() { return 256; }
#endif
  struct El { i32 v[9]; };
  // r = a*b: inputs |limb| <= ~1.8 * 2^29 product-magnitude <= 3.5; output limbs in [0, 2^29 + small)
  HD static El mul(const El& a, const El& b) {
    El r;
    i64 d, c;
    const i32 R1 = r1();
    // column 9
    d = 0;
#pragma unroll
    for (int i = 1; i <= 8; i++) d = smad(d, a.v[i], b.v[9 - i]);
    i32 h = (i32)((u32)d & M); d >>= 29;
    c = smad(0, a.v[0], b.v[0]);
    c = smadk(c, h, R0);
    r.v[0] = (i32)((u32)c & M); c >>= 29;
    i32 hp = h;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
      // d: column 9 + k
#pragma unroll
      for (int i = k + 1; i <= 8; i++) d = smad(d, a.v[i], b.v[9 + k - i]);
      if (k < 8) { h = (i32)((u32)d & M); d >>= 29; } else { h = (i32)d; }
#pragma unroll
      for (int i = 0; i <= k; i++) c = smad(c, a.v[i], b.v[k - i]);
      c = smadk(c, hp, R1);
      c = smadk(c, h, R0);
      r.v[k] = (i32)((u32)c & M); c >>= 29;
      hp = h;
    }
    // limb 9 value: c + h8*R1   (weight 2^261)
    c = smadk(c, hp, R1);
    i32 g = (i32)((u32)c & M); i32 g2 = (i32)(c >> 29);
    i64 t = smadk((i64)r.v[0], g, R0);
    r.v[0] = (i32)((u32)t & M); t >>= 29;
    t += (i64)r.v[1];
    t = smadk(t, g, R1);
    t = smadk(t, g2, R0);
    r.v[1] = (i32)((u32)t & M); t >>= 29;
    r.v[2] += (i32)t + g2 * R1;
    return r;
  }
  HD static El sqr(const El& a) {
    El r;
    const i32 R1 = r1();
    i32 a2[9];
#pragma unroll
    for (int i = 0; i < 9; i++) a2[i] = a.v[i] * 2;
    auto col = [&](int k, i64 s) -> i64 {
#pragma unroll
      for (int i = 0; i <= 8; i++) {
        int j = k - i;
        if (j < 0 || j > 8 || i > j) continue;
        s = (i == j) ? smad(s, a.v[i], a.v[i]) : smad(s, a.v[i], a2[j]);
      }
      return s;
    };
    i64 d = col(9, 0), c;
    i32 h = (i32)((u32)d & M); d >>= 29;
    c = col(0, 0);
    c = smadk(c, h, R0);
    r.v[0] = (i32)((u32)c & M); c >>= 29;
    i32 hp = h;
#pragma unroll
    for (int k = 1; k <= 8; k++) {
      if (k < 8) { d = col(9 + k, d); h = (i32)((u32)d & M); d >>= 29; } else { h = (i32)d; }
      c = col(k, c);
      c = smadk(c, hp, R1);
      c = smadk(c, h, R0);
      r.v[k] = (i32)((u32)c & M); c >>= 29;
      hp = h;
    }
    c = smadk(c, hp, R1);
    i32 g = (i32)((u32)c & M); i32 g2 = (i32)(c >> 29);
    i64 t = smadk((i64)r.v[0], g, R0);
    r.v[0] = (i32)((u32)t & M); t >>= 29;
    t += (i64)r.v[1];
    t = smadk(t, g, R1);
    t = smadk(t, g2, R0);
    r.v[1] = (i32)((u32)t & M); t >>= 29;
    r.v[2] += (i32)t + g2 * R1;
    return r;
  }
  HD static El add(const El& a, const El& b) { El r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
  HD static El sub(const El& a, const El& b) { El r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
  // parallel carry: limbs back to [-small, 2^29 + small)
  HD static El norm(const El& a) { El r; i32 c[9]; const i32 R1 = 256;
#pragma unroll
    for (int i = 0; i < 9; i++) c[i] = a.v[i] >> 29;
    r.v[0] = (i32)((u32)a.v[0] & M) + c[8] * R0;
    r.v[1] = (i32)((u32)a.v[1] & M) + c[0] + c[8] * R1;
#pragma unroll
    for (int i = 2; i < 9; i++) r.v[i] = (i32)((u32)a.v[i] & M) + c[i - 1];
    return r; }
};
