// ellgpu -- SHA-512 / SHA-384 of a three-part message per lane: EdDSA's hashInt
// (lib/elliptic/eddsa/index.js:65-70) and the HMAC-DRBG of EC#sign on p384 / p521
// (hash.js sha512 / sha384 are FIPS 180-4).  Hashing is ~1 % of an EdDSA verification, so this
// is a plain byte-wise implementation: one lane = one message, blocks assembled from byte loads.
#pragma once

#include "common.h"

namespace ell {

// OUT = 64: SHA-512; OUT = 48: SHA-384 (other initial state, truncated output)
template <int OUT_>
struct Sha2Wide {
  static constexpr int OUT = OUT_;
  static constexpr int BLOCK = 128;
  ELL_HD static u64 K(int i) {
    const u64 k[80] = {
        0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL,
        0x3956c25bf348b538ULL, 0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL,
        0xd807aa98a3030242ULL, 0x12835b0145706fbeULL, 0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL,
        0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL, 0xc19bf174cf692694ULL,
        0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
        0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL,
        0x983e5152ee66dfabULL, 0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL,
        0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL, 0x06ca6351e003826fULL, 0x142929670a0e6e70ULL,
        0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL, 0x53380d139d95b3dfULL,
        0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
        0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL,
        0xd192e819d6ef5218ULL, 0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL,
        0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL, 0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL,
        0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL, 0x682e6ff3d6b2b8a3ULL,
        0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
        0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL,
        0xca273eceea26619cULL, 0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL,
        0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL, 0x113f9804bef90daeULL, 0x1b710b35131c471bULL,
        0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL, 0x431d67c49c100d4cULL,
        0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};
    return k[i];
  }
  ELL_HD static u64 rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }

  // byte j of the virtual message  p0[0..l0) || p1[0..l1) || p2[0..l2) || 0x80 || 0.. || len
  ELL_HD static u32 msg_byte(u64 j, const u8* p0, u64 l0, const u8* p1, u64 l1, const u8* p2,
                             u64 l2, u64 total, u64 padded) {
    if (j < l0) return p0[j];
    if (j < l0 + l1) return p1[j - l0];
    if (j < total) return p2[j - l0 - l1];
    if (j == total) return 0x80u;
    if (j >= padded - 8) {                       // 128-bit big-endian bit length, low 64 bits
      u64 bits = total * 8;
      return (u32)((bits >> (8 * (padded - 1 - j))) & 0xffu);
    }
    return 0u;
  }

  // digest[0..OUT) = SHA-512 / SHA-384 (p0 || p1 || p2)
  ELL_HD static void hash3(u8 (&digest)[OUT], const u8* p0, u64 l0, const u8* p1, u64 l1,
                           const u8* p2, u64 l2) {
    u64 h[8];
    if (OUT == 64) {
      const u64 iv[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                         0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                         0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
      ELL_UNROLL
      for (int i = 0; i < 8; i++) h[i] = iv[i];
    } else {
      const u64 iv[8] = {0xcbbb9d5dc1059ed8ULL, 0x629a292a367cd507ULL, 0x9159015a3070dd17ULL,
                         0x152fecd8f70e5939ULL, 0x67332667ffc00b31ULL, 0x8eb44a8768581511ULL,
                         0xdb0c2e0d64f98fa7ULL, 0x47b5481dbefa4fa4ULL};
      ELL_UNROLL
      for (int i = 0; i < 8; i++) h[i] = iv[i];
    }
    const u64 total = l0 + l1 + l2;
    const u64 padded = ((total + 1 + 16 + 127) / 128) * 128;
    ELL_NOUNROLL
    for (u64 blk = 0; blk < padded; blk += 128) {
      u64 w[16];
      ELL_NOUNROLL
      for (int t = 0; t < 16; t++) {
        u64 x = 0;
        ELL_NOUNROLL
        for (int b = 0; b < 8; b++)
          x = (x << 8) | msg_byte(blk + 8 * t + b, p0, l0, p1, l1, p2, l2, total, padded);
        w[t] = x;
      }
      u64 a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
      ELL_NOUNROLL
      for (int r0 = 0; r0 < 80; r0 += 16) {
        ELL_UNROLL
        for (int i = 0; i < 16; i++) {
          const int t = r0 + i;
          u64 wt;
          if (r0 == 0) wt = w[i];
          else {
            u64 w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            u64 s0 = rotr(w15, 1) ^ rotr(w15, 8) ^ (w15 >> 7);
            u64 s1 = rotr(w2, 19) ^ rotr(w2, 61) ^ (w2 >> 6);
            wt = w[i] + s0 + w[(i + 9) & 15] + s1;
            w[i] = wt;
          }
          u64 S1 = rotr(e, 14) ^ rotr(e, 18) ^ rotr(e, 41);
          u64 ch = (e & f) ^ (~e & g);
          u64 t1 = hh + S1 + ch + K(t) + wt;
          u64 S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
          u64 mj = (a & b) ^ (a & c) ^ (b & c);
          u64 t2 = S0 + mj;
          hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
      }
      h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
    ELL_UNROLL
    for (int i = 0; i < OUT / 8; i++)
      ELL_UNROLL
      for (int b = 0; b < 8; b++) digest[8 * i + b] = (u8)(h[i] >> (56 - 8 * b));
  }
};
typedef Sha2Wide<64> Sha512;
typedef Sha2Wide<48> Sha384;

}  // namespace ell
