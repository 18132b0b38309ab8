// ellgpu -- HMAC_DRBG over SHA-384 / SHA-512, word-oriented (64-bit words): the register-resident
// form of hmac_drbg.h for p384, where n.byteLength() = 48 is a whole number of words and one V
// block covers a draw; HmacDrbg512Bytes below handles p521's 66-byte seeds and draws.  Layout and
// compression counts as in hmac_drbg256.h.  OUTW = digest words (6: SHA-384, 8: SHA-512),
// SEEDW = 64-bit words of entropy || nonce.
#pragma once

#include "common.h"
#include "sha512.h"

namespace ell {

template <int OUTW>
struct Sha512W {
  ELL_HD static u64 rotr(u64 x, int n) { return (x >> n) | (x << (64 - n)); }
  ELL_HD static void iv(u64 (&st)[8]) {
    if (OUTW == 8) {
      const u64 h[8] = {0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                        0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                        0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL};
      ELL_UNROLL
      for (int i = 0; i < 8; i++) st[i] = h[i];
    } else {
      const u64 h[8] = {0xcbbb9d5dc1059ed8ULL, 0x629a292a367cd507ULL, 0x9159015a3070dd17ULL,
                        0x152fecd8f70e5939ULL, 0x67332667ffc00b31ULL, 0x8eb44a8768581511ULL,
                        0xdb0c2e0d64f98fa7ULL, 0x47b5481dbefa4fa4ULL};
      ELL_UNROLL
      for (int i = 0; i < 8; i++) st[i] = h[i];
    }
  }
  // one compression of a 16-word (big-endian, 64-bit) block
  ELL_HD static void compress(u64 (&st)[8], const u64 (&blk)[16]) {
    u64 w[16];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    u64 a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], hh = st[7];
    ELL_NOUNROLL
    for (int r0 = 0; r0 < 80; r0 += 16) {
      ELL_UNROLL
      for (int i = 0; i < 16; i++) {
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
        u64 t1 = hh + S1 + ch + Sha512::K(r0 + i) + wt;
        u64 S0 = rotr(a, 28) ^ rotr(a, 34) ^ rotr(a, 39);
        u64 mj = (a & b) ^ (a & c) ^ (b & c);
        u64 t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
      }
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += hh;
  }
};

// SHA-512 of  prefix (PW 64-bit big-endian words held in registers) || msg[0..len): EdDSA's
// hashInt inputs (eddsa/index.js:65-70) are 0, 32 or 64 known bytes followed by the message.
// The prefix words and every whole message word are assembled without byte loops (the message
// is read with 8-byte loads + a byte swap); only the word holding the end of the message is put
// together byte by byte.
// `init` (optional) = the state after `prior` bytes (a whole number of blocks) already hashed:
// HMAC's inner hash continues from the state of the ipad block.
template <int PW>
ELL_HD void sha512_prefixed(u64 (&st)[8], const u64* pre, const u8* msg, u64 len,
                            const u64* init = nullptr, u64 prior = 0) {
  typedef Sha512W<8> H;
  static_assert(PW >= 0 && PW < 14, "prefix must leave room in the first block");
  if (init) {
    ELL_UNROLL
    for (int i = 0; i < 8; i++) st[i] = init[i];
  } else {
    H::iv(st);
  }
  const u64 total = 8ull * PW + len;
  const u64 nblk = (total + 1 + 16 + 127) / 128;
  // message word starting at byte `off` of msg (may lie beyond its end: padding)
  auto msg_word = [&](u64 off) -> u64 {
    if (off + 8 <= len) {
      u64 x;
      __builtin_memcpy(&x, msg + off, 8);
      return __builtin_bswap64(x);
    }
    u64 x = 0;
    ELL_UNROLL
    for (int b = 0; b < 8; b++) {
      u64 j = off + b;
      u64 v = j < len ? (u64)msg[j] : (j == len ? 0x80ull : 0ull);
      x = (x << 8) | v;
    }
    return x;
  };
  ELL_NOUNROLL
  for (u64 blk = 0; blk < nblk; blk++) {
    u64 w[16];
    const bool last = blk + 1 == nblk;
    ELL_UNROLL
    for (int t = 0; t < 16; t++) {
      if (t < PW) {
        // prefix words live in block 0 only
        w[t] = blk == 0 ? pre[t] : msg_word(blk * 128 + 8 * t - 8ull * PW);
      } else {
        w[t] = msg_word(blk * 128 + 8 * t - 8ull * PW);
      }
    }
    if (last) { w[14] = 0; w[15] = (prior + total) * 8; }   // 128-bit length; messages are < 2^61 bytes
    H::compress(st, w);
  }
}

template <int OUTW, int SEEDW>
struct HmacDrbg512 {
  typedef Sha512W<OUTW> H;
  u64 Kw[OUTW], Vw[OUTW];    // K and V as big-endian 64-bit words
  u64 si[8], so[8];          // hash states after the ipad / opad block of the current K

  ELL_HD void key_states() {
    u64 blk[16];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] = (i < OUTW ? Kw[i] : 0ull) ^ 0x3636363636363636ULL;
    H::iv(si);
    H::compress(si, blk);
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] ^= 0x3636363636363636ULL ^ 0x5c5c5c5c5c5c5c5cULL;
    H::iv(so);
    H::compress(so, blk);
  }
  // outer hash: opad block || inner digest (8 OUTW bytes)
  ELL_HD void outer(u64 (&out)[OUTW], const u64 (&inner)[8]) const {
    u64 blk[16];
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) blk[i] = inner[i];
    blk[OUTW] = 0x8000000000000000ULL;
    ELL_UNROLL
    for (int i = OUTW + 1; i < 15; i++) blk[i] = 0;
    blk[15] = (128 + 8 * OUTW) * 8;
    u64 st[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) st[i] = so[i];
    H::compress(st, blk);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) out[i] = st[i];
  }
  // out = HMAC(K, V)
  ELL_HD void hmac_v(u64 (&out)[OUTW]) const {
    u64 blk[16], in[8];
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) blk[i] = Vw[i];
    blk[OUTW] = 0x8000000000000000ULL;
    ELL_UNROLL
    for (int i = OUTW + 1; i < 15; i++) blk[i] = 0;
    blk[15] = (128 + 8 * OUTW) * 8;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) in[i] = si[i];
    H::compress(in, blk);
    outer(out, in);
  }
  // out = HMAC(K, V || sep || seed[0..T) ), T = 0 or SEEDW words
  template <int T>
  ELL_HD void hmac_v_sep(u64 (&out)[OUTW], u64 sep, const u64* seed) const {
    constexpr int NW = OUTW + T + 1;                    // message words incl. the one holding 0x80
    constexpr int NBLK = NW + 2 <= 16 ? 1 : 2;
    u64 m[16 * NBLK];
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) m[i] = Vw[i];
    u64 carry = sep & 0xffull;
    ELL_UNROLL
    for (int j = 0; j < T; j++) {
      m[OUTW + j] = (carry << 56) | (seed[j] >> 8);
      carry = seed[j] & 0xffull;
    }
    m[OUTW + T] = (carry << 56) | 0x0080000000000000ULL;
    ELL_UNROLL
    for (int i = NW; i < 16 * NBLK - 1; i++) m[i] = 0;
    m[16 * NBLK - 1] = (128 + 8 * OUTW + 1 + 8 * T) * 8;
    u64 in[8];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) in[i] = si[i];
    ELL_UNROLL
    for (int b = 0; b < NBLK; b++) {
      u64 blk[16];
      ELL_UNROLL
      for (int i = 0; i < 16; i++) blk[i] = m[16 * b + i];
      H::compress(in, blk);
    }
    outer(out, in);
  }
  template <int T>
  ELL_HD void update(const u64* seed) {
    u64 t[OUTW];
    hmac_v_sep<T>(t, 0x00u, seed);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) Kw[i] = t[i];
    key_states();
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) Vw[i] = t[i];
    if (T == 0) return;
    hmac_v_sep<T>(t, 0x01u, seed);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) Kw[i] = t[i];
    key_states();
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) Vw[i] = t[i];
  }
  ELL_HD void init(const u64 (&seed)[SEEDW]) {
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) { Kw[i] = 0; Vw[i] = 0x0101010101010101ULL; }
    key_states();
    update<SEEDW>(seed);
  }
  // generate(len <= 8 OUTW bytes) in two halves (hmac_drbg256.h): draw = the new V, reseed = the
  // _update() that ends generate, run by the caller only in front of a further draw
  ELL_HD void draw(u64 (&out)[OUTW]) {
    u64 t[OUTW];
    hmac_v(t);
    ELL_UNROLL
    for (int i = 0; i < OUTW; i++) { Vw[i] = t[i]; out[i] = t[i]; }
  }
  ELL_HD void reseed() { update<0>(nullptr); }
};

// HMAC_DRBG over SHA-512 for seeds and draws that are not whole words (p521: 66-byte entropy,
// nonce and draws): K, V and the two key states live in registers as 64-bit words, the
// sep || seed tail of an update is read from a byte buffer 8 bytes at a time.
struct HmacDrbg512Bytes {
  typedef Sha512W<8> H;
  u64 Kw[8], Vw[8], si[8], so[8];

  ELL_HD void key_states() {
    u64 blk[16];
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] = (i < 8 ? Kw[i] : 0ull) ^ 0x3636363636363636ULL;
    H::iv(si);
    H::compress(si, blk);
    ELL_UNROLL
    for (int i = 0; i < 16; i++) blk[i] ^= 0x3636363636363636ULL ^ 0x5c5c5c5c5c5c5c5cULL;
    H::iv(so);
    H::compress(so, blk);
  }
  // out = HMAC(K, V || tail[0..len))
  ELL_HD void hmac_v_tail(u64 (&out)[8], const u8* tail, u64 len) const {
    u64 in[8];
    sha512_prefixed<8>(in, Vw, tail, len, si, 128);
    u64 blk[16];
    ELL_UNROLL
    for (int i = 0; i < 8; i++) blk[i] = in[i];
    blk[8] = 0x8000000000000000ULL;
    ELL_UNROLL
    for (int i = 9; i < 15; i++) blk[i] = 0;
    blk[15] = (128 + 64) * 8;
    ELL_UNROLL
    for (int i = 0; i < 8; i++) out[i] = so[i];
    H::compress(out, blk);
  }
  // hmac-drbg.js:54-69 _update; buf[0] is the separator slot, buf[1..1+seedlen) the seed
  ELL_HD void update(u8* buf, u64 seedlen) {
    u64 t[8];
    buf[0] = 0x00;
    hmac_v_tail(t, buf, 1 + seedlen);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Kw[i] = t[i];
    key_states();
    hmac_v_tail(t, nullptr, 0);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Vw[i] = t[i];
    if (seedlen == 0) return;
    buf[0] = 0x01;
    hmac_v_tail(t, buf, 1 + seedlen);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Kw[i] = t[i];
    key_states();
    hmac_v_tail(t, nullptr, 0);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) Vw[i] = t[i];
  }
  ELL_HD void init(u8* buf, u64 seedlen) {
    ELL_UNROLL
    for (int i = 0; i < 8; i++) { Kw[i] = 0; Vw[i] = 0x0101010101010101ULL; }
    key_states();
    update(buf, seedlen);
  }
  // :91-113 generate(len), 64 < len <= 128: two V blocks, the first `len` bytes as big-endian words
  // (draw2 / reseed: the two halves of generate, hmac_drbg256.h)
  ELL_HD void draw2(u64 (&out)[16]) {
    u64 t[8];
    hmac_v_tail(t, nullptr, 0);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) { Vw[i] = t[i]; out[i] = t[i]; }
    hmac_v_tail(t, nullptr, 0);
    ELL_UNROLL
    for (int i = 0; i < 8; i++) { Vw[i] = t[i]; out[8 + i] = t[i]; }
  }
  ELL_HD void reseed(u8* sepbuf) { update(sepbuf, 0); }
};

}  // namespace ell
