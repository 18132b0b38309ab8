// tests/hostsim -- CPU-only unit-test build of the device code.
//
// TEST INFRASTRUCTURE.  This compiles elliptic_amd/csrc/*.h (the very headers
// the HIP kernels are made of) with g++ and runs each "thread" in a for-loop,
// so that kernel LOGIC (field arithmetic, group law, recodings, ladders, batch
// inversion, ECDSA pre/post-processing, the engine's chunking) can be checked
// against the oracle in a container without a GPU.  It exports the same C ABI
// as libellgpu.so plus a few white-box probes.  It is built only by the tests
// (tests/hostsim/build.py -> tests/hostsim/_build/libellgpu_hostsim.so), is
// never loaded by elliptic_amd, and is not a fallback for anything.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <thread>
#include <vector>

#include "../../elliptic_amd/csrc/engine.h"

namespace ell {
// launches per kernel name since the last hs_launches_reset (white-box probe: which form of an
// operation a batch was routed to)
void hs_note_launch(const char* name);
struct LoopBackend {
  void use_stream(void*) {}
  int use_stream_dev(void*) { return 0; }
  void* alloc(size_t bytes) { return malloc(bytes ? bytes : 1); }
  std::vector<unsigned char> pin_;
  void* pinned(size_t bytes) { if (pin_.size() < bytes) pin_.resize(bytes); return pin_.data(); }
  void free_(void* p) { free(p); }
  void h2d(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); }
  void d2h(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); }
  int sync() { return 0; }
  int device_index() const { return 0; }
  int compute_units() const { return 256; }
  void* own_stream() const { return nullptr; }
  void rt_upload(const RtField& f) { rt_host_block() = f; }
  void sync_all() {}
  void end_call(bool) {}
  // Engine::pipelined: tiny quantum so that ordinary test batches are cut into several chunks
  static size_t pipeline_quantum() { return 8; }
  void h2d_copy(void* d, const void* h, size_t bytes) { memcpy(d, h, bytes); }
  void d2h_copy(void* h, const void* d, size_t bytes) { memcpy(h, d, bytes); }
  void copies_before_compute() {}
  int mark_compute() { return 0; }
  void copy_after(int) {}
  void select_lane(int) {}
  int sync_lanes() { return 0; }
  // "threads" of a launch are independent, so the loop is split over the host
  // cores (only to keep the CPU test-suite short)
  template <class Fn>
  void launch(const Fn& f, size_t nthreads) {
    hs_note_launch(Fn::NAME);
    unsigned hw = std::thread::hardware_concurrency();
    size_t nw = hw ? hw : 1;
    if (nthreads < 64) nw = 1;
    std::vector<std::thread> pool;
    for (size_t w = 0; w < nw; w++) {
      size_t lo = nthreads * w / nw, hi = nthreads * (w + 1) / nw;
      auto body = [&f, lo, hi]() {
        signed char digits[Fn::DS_PER_LANE > 0 ? Fn::DS_PER_LANE : 1];
        DigitStore ds{digits, 1};
        for (size_t t = lo; t < hi; t++) f(t, ds);
      };
      if (nw == 1) body(); else pool.emplace_back(body);
    }
    for (auto& th : pool) th.join();
  }
  // the lanes-per-item layer: one unit per wave on the device, one call per unit here -- the
  // host build of coop.h holds all sixteen lanes of the row in every field element
  template <class Fn>
  void launch_coop(const Fn& f, size_t units) {
    hs_note_launch(Fn::NAME);
    unsigned hw = std::thread::hardware_concurrency();
    size_t nw = hw ? hw : 1;
    if (units < 8) nw = 1;
    if (nw > units) nw = units;
    std::vector<std::thread> pool;
    for (size_t w = 0; w < nw; w++) {
      size_t lo = units * w / nw, hi = units * (w + 1) / nw;
      auto body = [&f, lo, hi]() {
        signed char digits[Fn::DS_PER_LANE > 0 ? Fn::DS_PER_LANE : 1];
        alignas(16) unsigned char rows[Fn::ROW_BYTES];
        DigitStore ds{digits, 1};
        for (size_t t = lo; t < hi; t++) f(t, ds, (void*)rows);
      };
      if (nw == 1) body(); else pool.emplace_back(body);
    }
    for (auto& th : pool) th.join();
  }
};
}  // namespace ell

#define ELL_BACKEND ell::LoopBackend
static int ell_backend_create(int, ell::LoopBackend*, std::string*) { return 0; }
static void ell_backend_destroy(ell::LoopBackend*) {}
static int ell_device_count() { return 1; }

#include "../../elliptic_amd/csrc/capi_common.h"

// ---- white-box probes (hostsim only) -------------------------------------------
using namespace ell;

template <class F, class = void>
struct has_wide_probe { static constexpr bool value = false; };
template <class RR>
struct has_wide_probe<FpSolinas<RR>, void> { static constexpr bool value = true; };
template <class F>
static void field_op(int op, const u32* a, const u32* b, u32* r) {
  typename F::El x, y, z;
  u32 ta[F::L], tb[F::L], tr[F::L];
  for (int i = 0; i < F::L; i++) { ta[i] = a[i]; tb[i] = b[i]; }
  x = F::from_plain(ta);
  y = F::from_plain(tb);
  switch (op) {
    case 0: z = F::add(x, y); break;
    case 1: z = F::sub(x, y); break;
    case 2: z = F::mul(x, y); break;
    case 3: z = F::sqr(x); break;
    case 4: if constexpr (!std::is_same<F, FpK256C>::value) z = F::inv(x); else z = x; break;
    case 5: z = F::neg(x); break;
    case 6: z = F::template mul_pow2<1>(x); break;
    case 7: z = F::template mul_pow2<2>(x); break;
    case 8: z = F::template mul_pow2<3>(x); break;
    default: z = x;
  }
  if constexpr (has_wide_probe<F>::value) {
    if (op == 13) {                          // reduce_wide of the raw 2L words (a | b << 32L)
      u32 w[2 * F::L];
      for (int i = 0; i < F::L; i++) { w[i] = ta[i]; w[F::L + i] = tb[i]; }
      z = F::reduce_wide(w);
    }
  }
  if constexpr (has_wide_probe<F>::value) {
    // pair products with one fold (fp.h FpSolinas::mul_sub_mul / mul_sub_sqr8; whether a curve's group
    // law USES them is the prime's PAIR flag)
    if (op == 17) z = F::mul_sub_mul(x, y, F::sub(y, x), F::add(x, y));        // x y - (y - x)(x + y)
    if (op == 18) z = F::mul_sub_sqr8(x, y, F::sub(x, y));                      // x y - 8 (x - y)^2
  }
  if constexpr (std::is_same<F, FpK256L>::value || std::is_same<F, FpK256C>::value) {
    if (op == 11) z = F::mul2(x, y, F::template neg_l<4>(x), F::template sub_l<4>(x, y));
    if (op == 12) { typename F::El a = F::sqr(x); z = F::norm(F::add_l(a, F::half_l(a))); }
  }
  if constexpr (std::is_same<F, FpK256C>::value) {
    // the row layer's own corners: the exact zero test of a product, a lazy difference through a
    // product, an entry read from the one-lane tables' memory format
    if (op == 14) { z = F::mul(x, y); z = F::is_zero_w(z) == F::is_zero(z) ? z : F::one(); }
    if (op == 15) z = F::mul(F::template sub_l<4>(x, y), F::add_l(x, y));
    if (op == 16) z = F::norm(F::template cneg_l<2>(F::load_words(ta), (tb[0] & 1u) != 0));
  }
  F::to_plain(tr, z);
  for (int i = 0; i < F::L; i++) r[i] = tr[i];
}

#include <map>
#include <mutex>
namespace ell {
static std::map<std::string, int>& hs_launch_map() { static std::map<std::string, int> m; return m; }
static std::mutex hs_launch_mu;
void hs_note_launch(const char* name) {
  std::lock_guard<std::mutex> g(hs_launch_mu);
  hs_launch_map()[name]++;
}
}  // namespace ell

// EC#sign's nonce generators, several draws in a row (the reseed in front of a second draw is the
// path no real signature takes): kind 0 = HmacDrbg256 (seed = 2 * nbytes, nbytes in {24, 28, 32}),
// 1 = HmacDrbg512<6, 12> (SHA-384, nbytes 48), 2 = HmacDrbg512Bytes (SHA-512, nbytes 66),
// 3 = the generic byte-wise HmacDrbg<Sha256>.  out: ndraws * nbytes bytes.
template <int NW>
static void drbg256_draws(const u8* seed, int ndraws, u8* out) {
  u32 sw[2 * NW];
  for (int w = 0; w < 2 * NW; w++)
    sw[w] = ((u32)seed[4 * w] << 24) | ((u32)seed[4 * w + 1] << 16) | ((u32)seed[4 * w + 2] << 8) | seed[4 * w + 3];
  HmacDrbg256<2 * NW> g;
  g.init(sw);
  for (int d = 0; d < ndraws; d++) {
    u32 v[8];
    if (d) g.reseed();
    g.draw(v);
    for (int b = 0; b < 4 * NW; b++) out[d * 4 * NW + b] = (u8)(v[b >> 2] >> (24 - 8 * (b & 3)));
  }
}
extern "C" {
void hs_launches_reset() {
  std::lock_guard<std::mutex> g(ell::hs_launch_mu);
  ell::hs_launch_map().clear();
}
int hs_launches(const char* name) {
  std::lock_guard<std::mutex> g(ell::hs_launch_mu);
  auto it = ell::hs_launch_map().find(name);
  return it == ell::hs_launch_map().end() ? 0 : it->second;
}
// field: 0 k256, 1 25519, 2.. mont(curve p): 10+curve -> base field, 20+curve -> order field
int hs_field_limbs(int field) {
  switch (field) {
    case 0: case 1: case 2: case 3: return 8;
    case 10: return 8; case 11: return 6; case 12: return 7; case 13: return 8; case 14: return 12; case 15: return 17;
    case 31: return 6; case 32: return 7; case 33: return 8; case 34: return 12; case 35: return 17;
    case 20: return 8; case 21: return 6; case 22: return 7; case 23: return 8; case 24: return 12; case 25: return 17;
    case 26: return 8;
  }
  return -1;
}
int hs_field_op(int field, int op, const u32* a, const u32* b, u32* r) {
  switch (field) {
    case 0: field_op<FpK256>(op, a, b, r); break;
    case 2: field_op<FpK256L>(op, a, b, r); break;
    case 3: field_op<FpK256C>(op, a, b, r); break;
    // the row layer's Montgomery fields (csrc/coop_mont.h), host simulation of the row
    case 31: field_op<CoopNist<CvP192>::F>(op, a, b, r); break;
    case 32: field_op<CoopNist<CvP224>::F>(op, a, b, r); break;
    case 33: field_op<CoopNist<CvP256>::F>(op, a, b, r); break;
    // the WIDE fields (csrc/coop_wide.h: 28-bit digits over the lanes of a wave), host simulation of the wave
    case 34: field_op<CoopNist<CvP384>::F>(op, a, b, r); break;
    case 35: field_op<CoopNist<CvP521>::F>(op, a, b, r); break;
    case 1:
      if (op == 10) {                      // mul_u32 by the one-limb constant b[0]
        u32 ta[8];
        for (int i = 0; i < 8; i++) ta[i] = a[i];
        Fp25519::El z = Fp25519::mul_u32(Fp25519::from_plain(ta), b[0]);
        for (int i = 0; i < 8; i++) r[i] = z.v[i];
      } else {
        field_op<Fp25519>(op, a, b, r);
      }
      break;
    case 10: field_op<FpMont<consts::SECP256K1_P>>(op, a, b, r); break;
    case 11: field_op<CvP192::F>(op, a, b, r); break;
    case 12: field_op<CvP224::F>(op, a, b, r); break;
    case 13: field_op<CvP256::F>(op, a, b, r); break;
    case 14: field_op<CvP384::F>(op, a, b, r); break;
    case 15: field_op<CvP521::F>(op, a, b, r); break;
    case 20: field_op<CvSecp256k1::Fn>(op, a, b, r); break;
    case 21: field_op<CvP192::Fn>(op, a, b, r); break;
    case 22: field_op<CvP224::Fn>(op, a, b, r); break;
    case 23: field_op<CvP256::Fn>(op, a, b, r); break;
    case 24: field_op<CvP384::Fn>(op, a, b, r); break;
    case 25: field_op<CvP521::Fn>(op, a, b, r); break;
    case 26: field_op<FpMont<consts::ED25519_N>>(op, a, b, r); break;
    default: return -1;
  }
  return 0;
}
int hs_drbg_draws(int kind, const u8* seed, int nbytes, int ndraws, u8* out) {
  if (kind == 0 && nbytes == 24) { drbg256_draws<6>(seed, ndraws, out); return 0; }
  if (kind == 0 && nbytes == 28) { drbg256_draws<7>(seed, ndraws, out); return 0; }
  if (kind == 0 && nbytes == 32) { drbg256_draws<8>(seed, ndraws, out); return 0; }
  if (kind == 1 && nbytes == 48) {
    u64 sw[12];
    for (int w = 0; w < 12; w++) { sw[w] = 0; for (int b = 0; b < 8; b++) sw[w] = (sw[w] << 8) | seed[8 * w + b]; }
    HmacDrbg512<6, 12> g;
    g.init(sw);
    for (int d = 0; d < ndraws; d++) {
      u64 v[6];
      if (d) g.reseed();
      g.draw(v);
      for (int b = 0; b < 48; b++) out[d * 48 + b] = (u8)(v[b >> 3] >> (56 - 8 * (b & 7)));
    }
    return 0;
  }
  if (kind == 2 && nbytes == 66) {
    u8 sb[1 + 132];
    for (int b = 0; b < 132; b++) sb[1 + b] = seed[b];
    HmacDrbg512Bytes g;
    g.init(sb, 132);
    for (int d = 0; d < ndraws; d++) {
      u64 v[16];
      if (d) g.reseed(sb);
      g.draw2(v);
      for (int b = 0; b < 66; b++) out[d * 66 + b] = (u8)(v[b >> 3] >> (56 - 8 * (b & 7)));
    }
    return 0;
  }
  if (kind == 3 && nbytes <= 66) {
    HmacDrbg<Sha256> g;
    g.init(seed, nbytes, seed + nbytes, nbytes);
    for (int d = 0; d < ndraws; d++) {
      if (d) g.reseed();
      g.draw(out + d * nbytes, nbytes);
    }
    return 0;
  }
  return -1;
}
// GLV split of a 256-bit k (8 LE limbs): k1, k2 as 5 LE limbs + sign flags
void hs_glv_split(const u32* k, u32* k1, int* neg1, u32* k2, int* neg2) {
  u32 kk[8], a[5], b[5];
  bool n1, n2;
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  glv_split(kk, a, n1, b, n2);
  for (int i = 0; i < 5; i++) { k1[i] = a[i]; k2[i] = b[i]; }
  *neg1 = n1; *neg2 = n2;
}
// the same with both halves made odd by lattice vectors (what the secp256k1 ladders use)
void hs_glv_split_odd(const u32* k, u32* k1, int* neg1, u32* k2, int* neg2) {
  u32 kk[8], a[5], b[5];
  bool n1, n2;
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  glv_split<true>(kk, a, n1, b, n2);
  for (int i = 0; i < 5; i++) { k1[i] = a[i]; k2[i] = b[i]; }
  *neg1 = n1; *neg2 = n2;
}
// signed 4-bit recoding probe: NNIB = 64 with carry window (65 digits)
void hs_recode64(const u32* k, signed char* digits) {
  u32 kk[8];
  for (int i = 0; i < 8; i++) kk[i] = k[i];
  DigitStore ds{digits, 1};
  recode_w4<8, 64, true>(kk, ds, 0, 1);
}
}
