// ellgpu -- HMAC_DRBG as the reference's EC#sign uses it (lib/elliptic/ec/index.js:141-148 ->
// hmac-drbg 1.0.1 lib/hmac-drbg.js, pinned in the reference's package-lock.json; NIST
// SP 800-90A without prediction resistance, no personalisation string): one generator per
// lane, seeded with entropy || nonce.  H supplies OUT, BLOCK and hash3 (sha256.h / sha512.h).
#pragma once

#include "common.h"

namespace ell {

template <class H>
struct HmacDrbg {
  static constexpr int OUT = H::OUT;
  static constexpr int BLOCK = H::BLOCK;
  static constexpr int MAXSEED = 2 * 66;          // entropy || nonce, n.byteLength() bytes each
  u8 K[OUT], V[OUT];

  // out = HMAC(K, a || b)   (hash.js hmac.js: inner pad 0x36, outer pad 0x5c)
  ELL_HD void hmac(u8 (&out)[OUT], const u8* a, u64 la, const u8* b, u64 lb) const {
    u8 kb[BLOCK];
    ELL_NOUNROLL
    for (int i = 0; i < BLOCK; i++) kb[i] = (u8)((i < OUT ? K[i] : 0) ^ 0x36);
    u8 inner[OUT];
    H::hash3(inner, kb, BLOCK, a, la, b, lb);
    ELL_NOUNROLL
    for (int i = 0; i < BLOCK; i++) kb[i] = (u8)(kb[i] ^ 0x36 ^ 0x5c);
    H::hash3(out, kb, BLOCK, inner, OUT, nullptr, 0);
  }
  // hmac-drbg.js:54-69 _update
  ELL_HD void update(const u8* seed, int seedlen) {
    u8 buf[1 + MAXSEED];
    ELL_NOUNROLL
    for (int i = 0; i < seedlen; i++) buf[1 + i] = seed[i];
    u8 t[OUT];
    buf[0] = 0x00;
    hmac(t, V, OUT, buf, 1 + (u64)seedlen);
    ELL_NOUNROLL
    for (int i = 0; i < OUT; i++) K[i] = t[i];
    hmac(t, V, OUT, nullptr, 0);
    ELL_NOUNROLL
    for (int i = 0; i < OUT; i++) V[i] = t[i];
    if (seedlen == 0) return;
    buf[0] = 0x01;
    hmac(t, V, OUT, buf, 1 + (u64)seedlen);
    ELL_NOUNROLL
    for (int i = 0; i < OUT; i++) K[i] = t[i];
    hmac(t, V, OUT, nullptr, 0);
    ELL_NOUNROLL
    for (int i = 0; i < OUT; i++) V[i] = t[i];
  }
  // :37-48 _init with seed = entropy || nonce
  ELL_HD void init(const u8* entropy, int elen, const u8* nonce, int nlen) {
    u8 seed[MAXSEED];
    ELL_NOUNROLL
    for (int i = 0; i < elen; i++) seed[i] = entropy[i];
    ELL_NOUNROLL
    for (int i = 0; i < nlen; i++) seed[elen + i] = nonce[i];
    ELL_NOUNROLL
    for (int i = 0; i < OUT; i++) { K[i] = 0x00; V[i] = 0x01; }
    update(seed, elen + nlen);
  }
  // :91-113 generate(len) without additional input, in two halves (hmac_drbg256.h): draw = the
  // output blocks, reseed = the _update() that ends generate, run only in front of a further draw
  ELL_HD void draw(u8* out, int len) {
    int have = 0;
    ELL_NOUNROLL
    while (have < len) {
      u8 t[OUT];
      hmac(t, V, OUT, nullptr, 0);
      ELL_NOUNROLL
      for (int i = 0; i < OUT; i++) {
        V[i] = t[i];
        if (have + i < len) out[have + i] = t[i];
      }
      have += OUT;
    }
  }
  ELL_HD void reseed() { update(nullptr, 0); }
};

}  // namespace ell
