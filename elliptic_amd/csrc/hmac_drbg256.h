// ellgpu -- HMAC_DRBG over SHA-256, word-oriented: the generator EC#sign draws its nonces from
// on secp256k1 / p192 / p224 / p256 (hmac_drbg.h is the generic byte-wise form and documents the
// algorithm; this one keeps K, V, the two keyed SHA-256 states and every message block in
// registers -- no byte buffers, 16 compressions per signature instead of ~32 byte-gathered ones).
// SEEDW = words of entropy || nonce (2 * n.byteLength() / 4).
#pragma once

#include "common.h"

namespace ell {

struct Sha256W {
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
  ELL_HD static void iv(u32 (&st)[8]) {
    const u32 h[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au,
                      0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
    ELL_UNROLL
    for (int i = 0; i < 8; i++) st[i] = h[i];
  }
  // one compression of a 16-word (big-endian) block
  ELL_HD static void compress(u32 (&st)[8], const u32 (&blk)[16]) {
    u32 w[16];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    u32 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], hh = st[7];
    ELL_NOUNROLL
    for (int r0 = 0; r0 < 64; r0 += 16) {
      ELL_UNROLL
      for (int i = 0; i < 16; i++) {
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
        u32 t1 = hh + S1 + ch + K(r0 + i) + wt;
        u32 S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        u32 mj = (a & b) ^ (a & c) ^ (b & c);
        u32 t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
      }
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += hh;
  }
};

template <int SEEDW>
struct HmacDrbg256 {
  u32 Kw[8], Vw[8];          // K and V as big-endian words
  u32 si[8], so[8];          // SHA-256 states after the ipad / opad block of the current K

  ELL_HD void key_states() {
    u32 blk[16];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] = (i < 8 ? Kw[i] : 0u) ^ 0x36363636u;
    Sha256W::iv(si);
    Sha256W::compress(si, blk);
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] ^= 0x36363636u ^ 0x5c5c5c5cu;
    Sha256W::iv(so);
    Sha256W::compress(so, blk);
  }
  // outer hash: opad block || inner digest (32 bytes) -> 96 bytes in total
  ELL_HD void outer(u32 (&out)[8], const u32 (&inner)[8]) const {
    u32 blk[16];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) blk[i] = inner[i];
    blk[8] = 0x80000000u;
    ELL_UNROLL
    for (int i = 9; i < 15; i++) blk[i] = 0;
    blk[15] = 96 * 8;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) out[i] = so[i];
    Sha256W::compress(out, blk);
  }
  // out = HMAC(K, V)
  ELL_HD void hmac_v(u32 (&out)[8]) const {
    u32 blk[16], in[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) blk[i] = Vw[i];
    blk[8] = 0x80000000u;
    ELL_UNROLL
    for (int i = 9; i < 15; i++) blk[i] = 0;
    blk[15] = 96 * 8;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) in[i] = si[i];
    Sha256W::compress(in, blk);
    outer(out, in);
  }
  // out = HMAC(K, V || sep || seed[0..T) ), T = 0 or SEEDW words
  template <int T>
  ELL_HD void hmac_v_sep(u32 (&out)[8], u32 sep, const u32* seed) const {
    constexpr int NW = 8 + T + 1;                       // message words incl. the one holding 0x80
    constexpr int NBLK = NW + 2 <= 16 ? 1 : 2;
    u32 m[16 * NBLK];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) m[i] = Vw[i];
    u32 carry = sep & 0xffu;
    ELL_UNROLL
    for (int j = 0; j < T; j++) {
      m[8 + j] = (carry << 24) | (seed[j] >> 8);
      carry = seed[j] & 0xffu;
    }
    m[8 + T] = (carry << 24) | 0x00800000u;
    ELL_UNROLL
    for (int i = NW; i < 16 * NBLK - 1; i++) m[i] = 0;
    m[16 * NBLK - 1] = (64 + 33 + 4 * T) * 8;
    u32 in[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) in[i] = si[i];
    ELL_UNROLL
    for (int b = 0; b < NBLK; b++) {
      u32 blk[16];
      ELL_UNROLL
      for (int i = 0; i < 16; i++) blk[i] = m[16 * b + i];
      Sha256W::compress(in, blk);
    }
    outer(out, in);
  }
  // hmac-drbg.js:54-69 _update(seed) / _update()
  template <int T>
  ELL_HD void update(const u32* seed) {
    u32 t[8];
    hmac_v_sep<T>(t, 0x00u, seed);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Kw[i] = t[i];
    key_states();
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Vw[i] = t[i];
    if (T == 0) return;
    hmac_v_sep<T>(t, 0x01u, seed);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Kw[i] = t[i];
    key_states();
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Vw[i] = t[i];
  }
  // :37-48 _init, seed = entropy || nonce as SEEDW big-endian words
  // (the two keyed states of the all-zero K the generator starts from are constants: the SHA-256
  // states after a block of 0x36 / of 0x5c bytes -- tests/hostsim hs_drbg_draws checks them against
  // Python's hmac through whole draws)
  ELL_HD void init(const u32 (&seed)[SEEDW]) {
    const u32 zi[8] = {0xf454deadu, 0x9725214fu, 0x90daf2a0u, 0xdf1228eau, 0x64e5750fu, 0xa3924181u, 0x824a932bu, 0xf8e04e32u};
    const u32 zo[8] = {0xd385480fu, 0x7abb6477u, 0x37c9c538u, 0x5dd82467u, 0x8e043a72u, 0x753434b0u, 0xdeb82818u, 0x361d45a6u};
    ELL_UNROLL
    for (int i = 0; i < 8; i++) { Kw[i] = 0; Vw[i] = 0x01010101u; si[i] = zi[i]; so[i] = zo[i]; }
    update<SEEDW>(seed);
  }
  // :91-113 generate(len <= 32 bytes) in two halves.  draw: V = HMAC(K, V), the new V is the output;
  // reseed: the _update() that ends generate (:109) -- it only matters to the NEXT draw, so the
  // caller runs it in front of a second draw instead of behind every first one (EC#sign takes its
  // first candidate in all but ~2^-128 of the cases: 16 compressions per signature instead of 24)
  ELL_HD void draw(u32 (&out)[8]) {
    u32 t[8];
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) { Vw[i] = t[i]; out[i] = t[i]; }
  }
  ELL_HD void reseed() { update<0>(nullptr); }
};

}  // namespace ell
