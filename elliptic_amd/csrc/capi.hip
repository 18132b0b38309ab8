// ellgpu -- libellgpu.so: the C ABI over Engine<HipBackend> (gfx950 / MI355X),
// context creation, and the integer-VALU roofline probe.  The kernels themselves
// are instantiated per (curve, operation) in inst.hip; all curve arithmetic is in
// the headers; there is no vendor library and no CPU path behind this library.
#include <type_traits>
#include <vector>

#include "engine_extern.h"

namespace ell {

// ---- integer-VALU roofline probe ----------------------------------------------
// Each lane runs `iters` rounds of 16 independent 32x32+64->64 multiply-adds
// (or the comparison instruction mixes); nothing is loaded inside the loop.
template <int KIND>
__global__ void __launch_bounds__(256) k_probe(u32* out, int iters, u32 seed) {
  u32 a = seed + threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u;
  u64 acc[16];
#pragma unroll
  for (int j = 0; j < 16; j++) acc[j] = (u64)(a + j) << 7;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int j = 0; j < 16; j++) {
      if (KIND == 0) {
        acc[j] = (u64)a * b + acc[j];                       // v_mad_u64_u32
      } else if (KIND == 1) {
        u32 lo = a * (u32)acc[j];                           // v_mul_lo_u32
        u32 hi = __umulhi(b, (u32)acc[j]);                  // v_mul_hi_u32
        acc[j] = ((u64)hi << 32) | lo;
      } else if (KIND == 2) {
        u32 lo = __umul24((u32)acc[j], a) + b;   // v_mad_u32_u24
        acc[j] = ((u64)(u32)(acc[j] >> 32) << 32) | lo;
      } else {
        acc[j] = acc[j] + (((u64)b << 32) | a);             // v_add_co + v_addc
      }
    }
    a += 0x1234567u;
    b ^= a;
  }
  u64 s = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) s ^= acc[j];
  out[(size_t)blockIdx.x * 256 + threadIdx.x] = (u32)s ^ (u32)(s >> 32);
}

// ---- field-layer throughput probe: `iters` dependent secp256k1 field / group operations per
// lane, one wavefront per workgroup so that `blocks` sets the number of resident waves per
// SIMD (1024 blocks = 1 wave on each SIMD).  kind: 10 mul chain, 11 sqr chain, 12 two
// interleaved mul chains, 13 add+sub chain, 14 Jacobian doubling chain, 15 mixed-add chain.
template <int KIND>
__global__ void __launch_bounds__(64) k_probe_field(u32* out, int iters, u32 seed) {
  typedef CvSecp256k1::F F;                  // the field the secp256k1 kernels run on (ELL_K256_LAZY)
  typedef ShortOps<CvSecp256k1> G;
  u32 xs[8], ys[8], zs[8];
#pragma unroll
  for (int i = 0; i < 8; i++) {
    xs[i] = seed * (i + 1) + threadIdx.x * 2654435761u;
    ys[i] = (seed ^ 0x9E3779B9u) * (i + 3) + threadIdx.x;
    zs[i] = seed + i * 0x1234567u + threadIdx.x * 7u;
  }
  xs[7] &= 0x7FFFFFFFu; ys[7] &= 0x7FFFFFFFu; zs[7] &= 0x7FFFFFFFu;
  F::El x = F::from_plain(xs), y = F::from_plain(ys), z = F::from_plain(zs);
  G::J p;
  p.X = x; p.Y = y; p.Z = z;
  G::A q;
  q.x = y; q.y = z;
  bool inf = false;
  u32 acc = 0;
#pragma nounroll
  for (int it = 0; it < iters; it++) {
    if (KIND == 10) x = F::mul(x, y);
    else if (KIND == 11) x = F::sqr(x);
    else if (KIND == 12) { x = F::mul(x, y); z = F::mul(z, y); }
    else if (KIND == 13) { x = F::add(x, y); x = F::sub(x, z); }
    else if (KIND == 14) p = G::dbl(p);
    else if (KIND == 15) p = G::add_mixed_lean(p, q, inf, [&]() { return q; });
    else {
      // 16 / 17 / 18: wide product / wide square / reduction of the SATURATED field only
      FpK256::El sx, sy;
#pragma unroll
      for (int i = 0; i < 8; i++) { sx.v[i] = xs[i]; sy.v[i] = ys[i]; }
      u32 t[16];
      if (KIND == 16) fe_mul_wide<8>(t, sx.v, sy.v);
      else if (KIND == 17) fe_sqr_wide<8>(t, sx.v);
      else {
#pragma unroll
        for (int i = 0; i < 8; i++) { t[i] = sx.v[i]; t[i + 8] = sy.v[i]; }
        sx = FpK256::reduce_wide(t);
#pragma unroll
        for (int i = 0; i < 8; i++) t[i] = sx.v[i];
      }
#pragma unroll
      for (int i = 0; i < 8; i++) { xs[i] = t[i] ^ (KIND == 18 ? 0u : t[i + 8]); }
      xs[7] &= 0x7FFFFFFFu;
    }
  }
  constexpr int NS = sizeof(F::El) / 4;
#pragma unroll
  for (int i = 0; i < NS; i++) acc ^= x.v[i] ^ z.v[i] ^ p.X.v[i] ^ p.Y.v[i] ^ p.Z.v[i];
#pragma unroll
  for (int i = 0; i < 8; i++) acc ^= xs[i];
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc ^ (inf ? 1u : 0u);
}

// the same chains on the lanes-per-item layer (coop.h; one item per wave): kind 20 product chain,
// 24 Jacobian doublings, 25 mixed additions -- what ONE item's critical path pays per operation
// (PR: one item per ROW -- four items per wave, coop.h FpK256R: kinds 30 / 34 / 35)
template <int KIND, bool PR = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_probe_row(u32* out, int iters, u32 seed) {
  typedef FpK256CT<PR> F;
  typedef ShortOps<CvSecp256k1CT<PR>> G;
  u32 xs[8], ys[8], zs[8];
  const u32 item = PR ? blockIdx.x * 4u + (threadIdx.x >> 4) : blockIdx.x;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    xs[i] = seed * (i + 1) + item * 2654435761u;
    ys[i] = (seed ^ 0x9E3779B9u) * (i + 3) + item;
    zs[i] = seed + i * 0x1234567u + item * 7u;
  }
  xs[7] &= 0x7FFFFFFFu; ys[7] &= 0x7FFFFFFFu; zs[7] &= 0x7FFFFFFFu;
  typename F::El x = F::from_plain(xs), y = F::from_plain(ys), z = F::from_plain(zs);
  typename G::J p;
  p.X = x; p.Y = y; p.Z = z;
  typename G::A q;
  q.x = y; q.y = z;
  bool inf = false;
#pragma nounroll
  for (int it = 0; it < iters; it++) {
    if (KIND == 20) x = F::mul(x, y);
    else if (KIND == 24) p = G::dbl(p);
    else p = G::add_mixed_lean(p, q, inf, [&]() { return q; });
  }
  // the canonical words of the results (row 0's item of a PR = false wave; every row's of a PR wave),
  // so that the two forms can be compared: out[item * 32 + ...] = x, X, Y, Z
  u32 w[4][8];
  F::to_plain(w[0], x); F::to_plain(w[1], p.X); F::to_plain(w[2], p.Y); F::to_plain(w[3], p.Z);
  if ((threadIdx.x & 15u) == 0 && (PR || threadIdx.x == 0)) {
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int l = 0; l < 8; l++) out[(size_t)item * 32 + c * 8 + l] = w[c][l] ^ (inf && c == 3 ? 1u : 0u);
  }
}

// ... and on the WIDE layer (coop_wide.h: p384 / p521, an element over the lanes of the wave): kinds
// 40 / 44 / 45 (p384 product / a = -3 doubling / mixed addition) and 50 / 54 / 55 (p521)
template <class CV1, int KIND>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(1, 2))) k_probe_wide(u32* out, int iters, u32 seed) {
  typedef CoopNist<CV1> CW;
  typedef typename CW::F F;
  typedef typename CW::G G;
  constexpr int L = F::L;
  u32 xs[L], ys[L], zs[L];
#pragma unroll
  for (int i = 0; i < L; i++) {
    xs[i] = seed * (i + 1) + blockIdx.x * 2654435761u;
    ys[i] = (seed ^ 0x9E3779B9u) * (i + 3) + blockIdx.x;
    zs[i] = seed + i * 0x1234567u + blockIdx.x * 7u;
  }
  xs[L - 1] &= 0xFFu; ys[L - 1] &= 0xFFu; zs[L - 1] &= 0xFFu;
  typename F::El x = F::from_plain(xs), y = F::from_plain(ys), z = F::from_plain(zs);
  typename G::J p;
  p.X = x; p.Y = y; p.Z = z;
  typename G::A q;
  q.x = y; q.y = z;
  bool inf = false;
#pragma nounroll
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) x = F::mul(x, y);
    else if (KIND == 4) p = G::dbl(p);
    else p = G::add_mixed_lean(p, q, inf, [&]() { return q; });
  }
  u32 w[4][L];
  F::to_plain(w[0], x); F::to_plain(w[1], p.X); F::to_plain(w[2], p.Y); F::to_plain(w[3], p.Z);
  if (threadIdx.x == 0) {
    u32 acc = inf ? 1u : 0u;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int l = 0; l < L; l++) acc ^= w[c][l] * (u32)(c * 31 + l + 1);
    out[(size_t)blockIdx.x * 32] = acc;
  }
}

// ---- white-box probe: one field operation per lane (tests/test_gpu_field.py) ----
template <class F, class = void>
struct ell_has_wide_probe { static constexpr bool value = false; };
template <class RR>
struct ell_has_wide_probe<FpSolinas<RR>, void> { static constexpr bool value = true; };
template <class F>
__global__ void k_field_op(int op, size_t n, const u32* a, const u32* b, u32* r) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  u32 ta[F::L], tb[F::L], tr[F::L];
  for (int l = 0; l < F::L; l++) { ta[l] = a[i * F::L + l]; tb[l] = b[i * F::L + l]; }
  typename F::El x = F::from_plain(ta), y = F::from_plain(tb), z;
  switch (op) {
    case 0: z = F::add(x, y); break;
    case 1: z = F::sub(x, y); break;
    case 2: z = F::mul(x, y); break;
    case 3: z = F::sqr(x); break;
    case 4: z = F::inv(x); break;
    case 5: z = F::neg(x); break;
    case 6: z = F::template mul_pow2<1>(x); break;
    case 7: z = F::template mul_pow2<2>(x); break;
    case 8: z = F::template mul_pow2<3>(x); break;
    default: z = x; break;
  }
  if constexpr (std::is_same<F, Fp25519>::value) {
    if (op == 10) z = F::mul_u32(x, tb[0]);               // one-limb constant
  }
  if constexpr (ell_has_wide_probe<F>::value) {
    // the reduction itself on ANY 2L-word value (a = low half, b = high half, not reduced first):
    // directed tests of FpSolinas's rarely taken fold branches, which products reach too seldom
    if (op == 13) {
      u32 w[2 * F::L];
      for (int l = 0; l < F::L; l++) { w[l] = ta[l]; w[F::L + l] = tb[l]; }
      z = F::reduce_wide(w);
    }
  }
  if constexpr (std::is_same<F, FpK256L>::value) {
    // lazy forms through the generated asm: x*y + (4p - x)*(x - y + 4p), and (3 x^2) / 2
    if (op == 11) z = F::mul2(x, y, F::template neg_l<4>(x), F::template sub_l<4>(x, y));
    if (op == 12) { typename F::El a = F::sqr(x); z = F::norm(F::add_l(a, F::half_l(a))); }
  }
  F::to_plain(tr, z);
  for (int l = 0; l < F::L; l++) r[i * F::L + l] = tr[l];
}
template <class F>
static void run_field_op(HipBackend& bk, int op, size_t n, const u32* a, const u32* b, u32* r) {
  hipLaunchKernelGGL(k_field_op<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, bk.cur, op, n, a, b, r);
}

}  // namespace ell

#define ELL_BACKEND ell::HipBackend

static int ell_device_count() {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return -1; }
  return n;
}
static int ell_backend_create(int device, ell::HipBackend* bk, std::string* err) {
  int n = ell_device_count();
  if (n <= 0) { *err = "no HIP device visible (libellgpu has no CPU fallback)"; return ell::E_NODEVICE; }
  if (device < 0 || device >= n) { *err = "device index out of range"; return ell::E_ARG; }
  if (hipSetDevice(device) != hipSuccess) { *err = "hipSetDevice failed"; return ell::E_NODEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { *err = "hipGetDeviceProperties failed"; return ell::E_NODEVICE; }
  std::string arch = prop.gcnArchName;
  if (arch.rfind("gfx950", 0) != 0) {
    *err = "device is " + arch + ", libellgpu is built for gfx950 (MI355X) only";
    return ell::E_NODEVICE;
  }
  bk->device = device;
  if (prop.multiProcessorCount > 0) bk->cus = prop.multiProcessorCount;
  bk->one_wave_groups = (size_t)bk->cus * 4 * 64 * 4;
  if (const char* e = getenv("ELLGPU_ONE_WAVE_GROUPS")) bk->one_wave_groups = (size_t)strtoull(e, nullptr, 10);
  if (hipStreamCreateWithFlags(&bk->own, hipStreamNonBlocking) != hipSuccess) { *err = "hipStreamCreate failed"; return ell::E_HIP; }
  bk->cur = bk->own;
  if (hipStreamCreateWithFlags(&bk->copy, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&bk->copy_out, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&bk->own2, hipStreamNonBlocking) != hipSuccess) { *err = "hipStreamCreate failed"; return ell::E_HIP; }
  for (int i = 0; i < ell::HipBackend::RING; i++)
    if (hipEventCreateWithFlags(&bk->ring[i], hipEventDisableTiming) != hipSuccess) { *err = "hipEventCreate failed"; return ell::E_HIP; }
  for (auto& l : bk->dlane)
    if (hipEventCreateWithFlags(&l.done, hipEventDisableTiming) != hipSuccess) { *err = "hipEventCreate failed"; return ell::E_HIP; }
  if (const char* e = getenv("ELLGPU_DEV_LANES")) bk->one_dev_lane = atoi(e) == 1;
  bk->timed = new std::vector<ell::TimedLaunch>();
  return ell::E_OK;
}
static void ell_backend_destroy(ell::HipBackend* bk) {
  if (bk->own) (void)hipStreamDestroy(bk->own);
  bk->own = nullptr;
  if (bk->copy) (void)hipStreamDestroy(bk->copy);
  if (bk->copy_out) (void)hipStreamDestroy(bk->copy_out);
  bk->copy_out = nullptr;
  if (bk->own2) (void)hipStreamDestroy(bk->own2);
  bk->copy = bk->own2 = nullptr;
  for (int i = 0; i < ell::HipBackend::RING; i++)
    if (bk->ring[i]) { (void)hipEventDestroy(bk->ring[i]); bk->ring[i] = nullptr; }
  if (bk->pin_buf) { (void)hipHostFree(bk->pin_buf); bk->pin_buf = nullptr; bk->pin_cap = 0; }
  for (auto& l : bk->dlane) {
    if (l.done) { (void)hipEventDestroy(l.done); l.done = nullptr; }
    l.stream = nullptr;
    l.pending = false;
  }
  if (bk->timed) {
    for (auto& t : *bk->timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
    delete bk->timed;
    bk->timed = nullptr;
  }
}

#include "capi_common.h"

extern "C" int ellgpu_probe_valu(ellgpu_ctx* ctx, int kind, int blocks, int iters, double* ms_out,
                                 double* ops_out) {
  ELL_ENTER(ctx, nullptr);
  if (blocks <= 0 || iters <= 0 || !ms_out || !ops_out) return set_err(ELLGPU_E_ARG, "bad probe arguments");
  ell::HipBackend& bk = ctx->eng->bk;
  ell::u32* out = (ell::u32*)bk.alloc((size_t)blocks * 256 * 4);
  if (!out) return set_err(ELLGPU_E_NOMEM, "probe allocation failed");
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int rep = 0; rep < 2; rep++) {           // first pass warms up clocks / code
    (void)hipEventRecord(e0, bk.cur);
    switch (kind) {
      case 0: hipLaunchKernelGGL(ell::k_probe<0>, dim3(blocks), dim3(256), 0, bk.cur, out, iters, 12345u); break;
      case 1: hipLaunchKernelGGL(ell::k_probe<1>, dim3(blocks), dim3(256), 0, bk.cur, out, iters, 12345u); break;
      case 2: hipLaunchKernelGGL(ell::k_probe<2>, dim3(blocks), dim3(256), 0, bk.cur, out, iters, 12345u); break;
      case 3: hipLaunchKernelGGL(ell::k_probe<3>, dim3(blocks), dim3(256), 0, bk.cur, out, iters, 12345u); break;
      case 10: hipLaunchKernelGGL(ell::k_probe_field<10>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 11: hipLaunchKernelGGL(ell::k_probe_field<11>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 12: hipLaunchKernelGGL(ell::k_probe_field<12>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 13: hipLaunchKernelGGL(ell::k_probe_field<13>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 14: hipLaunchKernelGGL(ell::k_probe_field<14>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 15: hipLaunchKernelGGL(ell::k_probe_field<15>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 16: hipLaunchKernelGGL(ell::k_probe_field<16>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 17: hipLaunchKernelGGL(ell::k_probe_field<17>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 18: hipLaunchKernelGGL(ell::k_probe_field<18>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 20: hipLaunchKernelGGL(ell::k_probe_row<20>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 24: hipLaunchKernelGGL(ell::k_probe_row<24>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 25: hipLaunchKernelGGL(ell::k_probe_row<25>, dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 40: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP384, 0>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 44: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP384, 4>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 45: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP384, 5>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 50: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP521, 0>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 54: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP521, 4>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 55: hipLaunchKernelGGL((ell::k_probe_wide<ell::CvP521, 5>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 30: hipLaunchKernelGGL((ell::k_probe_row<20, true>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 34: hipLaunchKernelGGL((ell::k_probe_row<24, true>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      case 35: hipLaunchKernelGGL((ell::k_probe_row<25, true>), dim3(blocks), dim3(64), 0, bk.cur, out, iters, 12345u); break;
      default: bk.free_(out); (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
               return set_err(ELLGPU_E_ARG, "unknown probe kind");
    }
    (void)hipEventRecord(e1, bk.cur);
    (void)hipEventSynchronize(e1);
  }
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  // the row kinds leave their results (k_probe_row): with ELLGPU_PROBE_DIGEST set, a 53-bit digest of
  // the first `blocks` ITEMS' words comes back in ops_out's place -- equal for kinds 2x and 3x when
  // the one-item-per-row field computes what the one-item-per-wave field computes
  double digest = -1.0;
  if (kind >= 20 && getenv("ELLGPU_PROBE_DIGEST")) {
    std::vector<ell::u32> h((size_t)blocks * (kind >= 30 ? 128 : 32));     // 32 words per item, four items per PR wave
    (void)hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long d = 1469598103934665603ull;
    for (ell::u32 v : h) { d ^= v; d *= 1099511628211ull; }
    digest = (double)(d >> 11);
  }
  bk.free_(out);
  *ms_out = ms;
  *ops_out = digest >= 0 ? digest : kind < 10 ? (double)blocks * 256.0 * (double)iters * 16.0
                       : (double)blocks * 64.0 * (double)iters * (kind == 12 || kind == 13 ? 2.0 : 1.0);
  return finish(ctx, bk.sync());
}

// Per-kernel timing (HIP events recorded around every launch on the stream the
// kernel runs on).  ellgpu_ctx_set_timing(ctx, 1) starts a fresh recording;
// ellgpu_ctx_get_timing synchronises and writes one line per kernel name:
// "<name> <launches> <total_ms>\n".  Returns the number of bytes written.
extern "C" int ellgpu_ctx_set_timing(ellgpu_ctx* ctx, int on) {
  if (!ctx) return set_err(ELLGPU_E_ARG, "null context");
  ELL_LOCK(ctx);                                   // launch() appends to bk.timed under the same lock
  ell::HipBackend& bk = ctx->eng->bk;
  for (auto& t : *bk.timed) { (void)hipEventDestroy(t.e0); (void)hipEventDestroy(t.e1); }
  bk.timed->clear();
  bk.timing = on != 0;
  return ELLGPU_OK;
}
extern "C" int ellgpu_ctx_get_timing(ellgpu_ctx* ctx, char* buf, size_t cap) {
  if (!ctx || !buf || !cap) return set_err(ELLGPU_E_ARG, "bad arguments");
  ELL_LOCK(ctx);
  ell::HipBackend& bk = ctx->eng->bk;
  (void)hipDeviceSynchronize();
  std::map<std::string, std::pair<int, double>> acc;
  for (auto& t : *bk.timed) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.e0, t.e1) != hipSuccess) { (void)hipGetLastError(); continue; }
    auto& a = acc[t.name];
    a.first += 1;
    a.second += ms;
  }
  std::string out;
  for (auto& kv : acc) {
    char line[160];
    snprintf(line, sizeof line, "%s %d %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (out.size() + 1 > cap) return set_err(ELLGPU_E_ARG, "timing buffer too small");
  memcpy(buf, out.c_str(), out.size() + 1);
  return (int)out.size();
}

// White-box probe (tests only): r[i] = a[i] <op> b[i] in one of the engine's fields, n items
// of `limbs` 32-bit little-endian limbs each (host pointers).  field: 0 secp256k1 p,
// 1 2^255-19, 10+c base field of short curve c as FpMont, 20+c order field of curve c
// (26 = ed25519 order).  op: 0 add, 1 sub, 2 mul, 3 sqr, 4 inv, 5 neg.
extern "C" int ellgpu_debug_field_op(ellgpu_ctx* ctx, int field, int op, size_t n, const uint32_t* a,
                                     const uint32_t* b, uint32_t* r) {
  using namespace ell;
  ELL_ENTER(ctx, nullptr);
  int L = 0;
  switch (field) {
    case 0: case 1: case 2: case 10: case 13: case 20: case 23: case 26: L = 8; break;
    case 11: case 21: L = 6; break;
    case 12: case 22: L = 7; break;
    case 14: case 24: L = 12; break;
    case 15: case 25: L = 17; break;
    default: return set_err(ELLGPU_E_ARG, "unknown field id");
  }
  HipBackend& bk = ctx->eng->bk;
  size_t bytes = n * (size_t)L * 4;
  u32* da = (u32*)bk.alloc(bytes); u32* db = (u32*)bk.alloc(bytes); u32* dr = (u32*)bk.alloc(bytes);
  if (!da || !db || !dr) return set_err(ELLGPU_E_NOMEM, "probe allocation failed");
  bk.h2d(da, a, bytes); bk.h2d(db, b, bytes);
  switch (field) {
    case 0: run_field_op<FpK256>(bk, op, n, da, db, dr); break;
    case 1: run_field_op<Fp25519>(bk, op, n, da, db, dr); break;
    case 2: run_field_op<FpK256L>(bk, op, n, da, db, dr); break;      // the 9 x 29-bit field (op 11: mul2, op 12: half)
    case 10: run_field_op<FpMont<consts::SECP256K1_P>>(bk, op, n, da, db, dr); break;
    case 11: run_field_op<CvP192::F>(bk, op, n, da, db, dr); break;
    case 12: run_field_op<CvP224::F>(bk, op, n, da, db, dr); break;
    case 13: run_field_op<CvP256::F>(bk, op, n, da, db, dr); break;
    case 14: run_field_op<CvP384::F>(bk, op, n, da, db, dr); break;
    case 15: run_field_op<CvP521::F>(bk, op, n, da, db, dr); break;
    case 20: run_field_op<CvSecp256k1::Fn>(bk, op, n, da, db, dr); break;
    case 21: run_field_op<CvP192::Fn>(bk, op, n, da, db, dr); break;
    case 22: run_field_op<CvP224::Fn>(bk, op, n, da, db, dr); break;
    case 23: run_field_op<CvP256::Fn>(bk, op, n, da, db, dr); break;
    case 24: run_field_op<CvP384::Fn>(bk, op, n, da, db, dr); break;
    case 25: run_field_op<CvP521::Fn>(bk, op, n, da, db, dr); break;
    case 26: run_field_op<FpMont<consts::ED25519_N>>(bk, op, n, da, db, dr); break;
  }
  bk.note(hipGetLastError());
  bk.d2h(r, dr, bytes);
  int rc = bk.sync();
  bk.free_(da); bk.free_(db); bk.free_(dr);
  return finish(ctx, rc);
}
