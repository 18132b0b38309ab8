// ellgpu -- SHA-256 of a three-part message per lane (hash.js sha256, FIPS 180-4): the hash
// of the HMAC-DRBG that EC#sign draws its nonces from on secp256k1 / p192 / p224 / p256
// (lib/elliptic/curves.js `hash:`).  Byte-wise like sha512.h: signing is not the hot path.
#pragma once

#include "common.h"

namespace ell {

struct Sha256 {
  static constexpr int OUT = 32;
  static constexpr int BLOCK = 64;
  ELL_HD static u32 K(int i) {
    const u32 k[64] = {
        0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u,
        0xd807aa98u, 0x12835b01u, 0x243185beu, 0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u,
        0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau, 0x5cb0a9dcu, 0x76f988dau,
        0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u,
        0x27b70a85u, 0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u,
        0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u, 0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u,
        0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu, 0x682e6ff3u,
        0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
    return k[i];
  }
  ELL_HD static u32 rotr(u32 x, int n) { return (x >> n) | (x << (32 - n)); }

  // byte j of the virtual message  p0 || p1 || p2 || 0x80 || 0.. || 64-bit big-endian bit length
  ELL_HD static u32 msg_byte(u64 j, const u8* p0, u64 l0, const u8* p1, u64 l1, const u8* p2,
                             u64 total, u64 padded) {
    if (j < l0) return p0[j];
    if (j < l0 + l1) return p1[j - l0];
    if (j < total) return p2[j - l0 - l1];
    if (j == total) return 0x80u;
    if (j >= padded - 8) return (u32)(((total * 8) >> (8 * (padded - 1 - j))) & 0xffu);
    return 0u;
  }

  // digest[0..32) = SHA-256(p0 || p1 || p2)
  ELL_HD static void hash3(u8 (&digest)[32], const u8* p0, u64 l0, const u8* p1, u64 l1,
                           const u8* p2, u64 l2) {
    u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    const u64 total = l0 + l1 + l2;
    const u64 padded = ((total + 1 + 8 + 63) / 64) * 64;
    ELL_NOUNROLL
    for (u64 blk = 0; blk < padded; blk += 64) {
      u32 w[16];
      ELL_NOUNROLL
      for (int t = 0; t < 16; t++) {
        u32 x = 0;
        ELL_NOUNROLL
        for (int b = 0; b < 4; b++) x = (x << 8) | msg_byte(blk + 4 * t + b, p0, l0, p1, l1, p2, total, padded);
        w[t] = x;
      }
      u32 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
      ELL_NOUNROLL
      for (int r0 = 0; r0 < 64; r0 += 16) {
        ELL_UNROLL
        for (int i = 0; i < 16; i++) {
          const int t = r0 + i;
          u32 wt;
          if (r0 == 0) wt = w[i];
          else {
            u32 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            u32 s0 = rotr(w15, 7) ^ rotr(w15, 18) ^ (w15 >> 3);
            u32 s1 = rotr(w2, 17) ^ rotr(w2, 19) ^ (w2 >> 10);
            wt = w[i] + s0 + w[(i + 9) & 15] + s1;
            w[i] = wt;
          }
          u32 S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
          u32 ch = (e & f) ^ (~e & g);
          u32 t1 = hh + S1 + ch + K(t) + wt;
          u32 S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
          u32 mj = (a & b) ^ (a & c) ^ (b & c);
          u32 t2 = S0 + mj;
          hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
      }
      h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    ELL_UNROLL
    for (int i = 0; i < 8; i++)
      ELL_UNROLL
      for (int b = 0; b < 4; b++) digest[4 * i + b] = (u8)(h[i] >> (24 - 8 * b));
  }
};

}  // namespace ell
