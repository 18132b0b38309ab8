// Micro-benchmark for a LANES-PER-ITEM field layer (round-4 review, "Next round" 2, step 1): one
// secp256k1 field element spread over the lanes of ONE 16-lane DPP row -- limb l (29 bits, signed,
// the radix of csrc/fpk256l.h) in lane l, lanes 9..15 zero -- against the one-item-per-lane
// multiply the kernels use today (ellgpu_probe_valu kinds 10 / 11 / 14, measured by the driver
// tools/microbench/coop_field.py in the same gpurun call).
//
// A lone EC#verify / Point#mul is ONE dependent chain of field operations and leaves the machine
// idle beside it; an instruction costs a wave its ~4.3 SIMD cycles whether 1 or 64 lanes are
// active.  So what counts for the single call is INSTRUCTIONS PER FIELD OPERATION on the critical
// path: the one-lane product is 72 v_mad_u64_u32 + ~100 carry / move instructions, a field
// addition a 13-instruction carry chain.  Here the nine partial-product rows of a product are
// nine v_mad_i64_i32 of ALL lanes at once (column l accumulates in lane l; operand a's limbs come
// as SGPRs through v_readlane, operand b shifted along the row by DPP row_shr), carries travel one
// lane up by DPP, the high half folds back through 2^261 = 256 * 2^29 + 31264 (mod p), and an
// addition or subtraction is ONE instruction.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o coop_field coop_field.hip
//   ./coop_field            -> JSON: ns per operation in a dependent chain on a lone wave
//                              + the limbs of a few results for the driver's big-integer check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int32_t i32;
typedef int64_t i64;
typedef uint32_t u32;
typedef uint64_t u64;

#define DEV __device__ __forceinline__
static constexpr i32 M29 = (1 << 29) - 1;
static constexpr i32 R0 = 31264, R1 = 256;            // 2^261 = R1 * 2^29 + R0 (mod p)

// lane l <- lane l - N of its 16-lane row, zero shifted in (DPP row_shr:N, bound_ctrl:0)
template <int N>
DEV i32 row_up(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + N, 0xF, 0xF, true); }
// lane l <- lane l + N, zero shifted in (row_shl:N)
template <int N>
DEV i32 row_down(i32 v) { return __builtin_amdgcn_update_dpp(0, v, 0x100 + N, 0xF, 0xF, true); }
DEV i32 lane_of(i32 v, int l) { return __builtin_amdgcn_readlane(v, l); }
// lane LANE of v <- the wave-uniform value s (v_writelane_b32)
template <int LANE>
DEV i32 set_lane(i32 s, i32 v) {
#if __has_builtin(__builtin_amdgcn_writelane)
  return __builtin_amdgcn_writelane(s, LANE, v);
#else
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(LANE));
  return v;
#endif
}
DEV i64 mad(i32 a, i32 b, i64 c) { return (i64)a * (i64)b + c; }        // v_mad_i64_i32

// One field element: this lane's limb.  Lanes 0..8 carry limbs, lanes 9..15 hold 0.
// N form, a little wider than fpk256l.h's: |limb| <= 2^29 + 2^24 (l < 8), limb 8 in (-2^5, 2^25 + 2^5).
struct CF { i32 v; };

// per-lane constants of the row (loop-invariant registers)
struct Lane {
  int lane;
  i32 mk;        // low-bits mask of a carry pass: 2^29 - 1, limb 8 keeps all its bits
  i32 kf;        // what one unit of limb 8's bit 24 is worth at this limb: 977, 8, 0 ... (2^256 = 2^32 + 977)
  i32 r1;        // R1 for lanes 1..8, 0 for lane 0 (the fold of column 9 + j onto limb j + 1)
  i32 rr;        // column 9 onto limbs 0 / 1: R0, R1, 0 ...
  i32 live;      // -1 for lanes 0..8, 0 above
};
DEV Lane make_lane(int lane) {
  Lane L;
  L.lane = lane;
  L.mk = lane == 8 ? -1 : M29;
  L.kf = lane == 0 ? 977 : (lane == 1 ? 8 : 0);
  L.r1 = lane == 0 ? 0 : R1;
  L.rr = lane == 0 ? R0 : (lane == 1 ? R1 : 0);
  L.live = lane <= 8 ? -1 : 0;
  return L;
}

DEV CF cadd(CF a, CF b) { return CF{a.v + b.v}; }                       // lazy: one instruction
DEV CF csub_raw(CF a, CF b) { return CF{a.v - b.v}; }                   // signed limbs: one instruction

// carry pass over 32-bit lazy limbs (|v| < 2^31, value >= 0): fpk256l.h norm() across the row
DEV CF cnorm(CF a, const Lane& L) {
  i32 c = a.v >> 29;
  i32 a8 = lane_of(a.v, 8);                           // limb 8's bits above 2^24 fold through 2^256
  i32 hi = a8 >> 24;
  i32 f = hi >= 1 ? hi - 1 : 0;                       // never all of it (fpk256l.h top_fold)
  i32 r = (a.v & L.mk) + row_up<1>(c);
  r -= L.lane == 8 ? (f << 24) : 0;
  r += f * L.kf;                                      // f < 2^7, kf < 2^10
  return CF{r & L.live};
}

// a * b mod p; column sums below 2^63 (N x N, or one operand a sum / difference of two N values);
// result in N form.
DEV CF cmul(CF a, CF b, const Lane& L) {
  // ---- columns 0..15, one per lane: nine multiply-accumulates of the whole row ----
  const i32 a0 = lane_of(a.v, 0), a1 = lane_of(a.v, 1), a2 = lane_of(a.v, 2), a3 = lane_of(a.v, 3),
            a4 = lane_of(a.v, 4), a5 = lane_of(a.v, 5), a6 = lane_of(a.v, 6), a7 = lane_of(a.v, 7),
            a8 = lane_of(a.v, 8);
  i64 acc = mad(a0, b.v, 0);
  acc = mad(a1, row_up<1>(b.v), acc);
  acc = mad(a2, row_up<2>(b.v), acc);
  acc = mad(a3, row_up<3>(b.v), acc);
  acc = mad(a4, row_up<4>(b.v), acc);
  acc = mad(a5, row_up<5>(b.v), acc);
  acc = mad(a6, row_up<6>(b.v), acc);
  acc = mad(a7, row_up<7>(b.v), acc);
  acc = mad(a8, row_up<8>(b.v), acc);
  // column 16 = a8 * b8 falls off the row: scalar unit (|.| < 2^52.4)
  i64 col16 = (i64)a8 * (i64)lane_of(b.v, 8);
  // ---- first carry pass (64-bit carries): limbs below 2^29 + 2^35 ----
  i64 c1 = acc >> 29;
  i32 c1l = (i32)c1, c1h = (i32)(c1 >> 32);
  i64 v1 = (i64)((i32)acc & M29) + (i64)(((u64)(u32)row_up<1>(c1h) << 32) | (u32)row_up<1>(c1l));
  col16 += (i64)(((u64)(u32)lane_of(c1h, 15) << 32) | (u32)lane_of(c1l, 15));
  // ---- second pass (carries below 2^6): 32-bit limbs below 2^29 + 2^6 ----
  i32 c2 = (i32)(v1 >> 29);
  i32 v2 = ((i32)v1 & M29) + row_up<1>(c2);
  col16 += (i64)lane_of(c2, 15);
  // columns 16 and 17.. as digits: H[7] in [0, 2^29), H[8] signed below 2^24
  const i32 p16 = (i32)col16 & M29;
  const i32 p17 = (i32)(col16 >> 29);
  // ---- fold: column 9 + j -> R0 at limb j, R1 at limb j + 1 (2^261 = R1 2^29 + R0) ----
  i32 h0 = row_down<9>(v2);                           // lane j <- column 9 + j   (j <= 6)
  h0 = set_lane<7>(p16, h0);
  h0 = set_lane<8>(p17, h0);
  i32 h1 = row_down<8>(v2);                           // lane j <- column 8 + j = H[j - 1]   (lane 0: times 0)
  h1 = set_lane<8>(p16, h1);
  const i32 w9 = p17 * R1;                            // column 17's R1 part lands on column 9: once more
  i64 t = mad(h0, R0, (i64)v2);
  t = mad(h1, L.r1, t);
  t = mad(w9, L.rr, t);
  // ---- carry pass over the folded limbs (below 2^46; the carries fit a word) ----
  i32 c3 = (i32)(t >> 29);
  i32 v3 = ((i32)t & M29) + row_up<1>(c3);
  // limb 8: its carry is column 9 again (small now), its bits above 2^24 fold through 2^256
  const i32 c9 = lane_of(c3, 8);
  const i32 x8 = lane_of(v3, 8);
  const i32 hi = x8 >> 24;
  i32 r = v3 + c9 * L.rr + hi * L.kf;                 // c9 < 2^9, hi < 2^6: small multiplies
  r -= L.lane == 8 ? (hi << 24) : 0;
  return CF{r & L.live};
}
DEV CF csqr(CF a, const Lane& L) { return cmul(a, a, L); }

// a - b + 4p in N form (the generic sub of fpk256l.h): limb 0 -= 4 * 977, limb 1 -= 32, limb 8 += 4 << 24
DEV CF csub(CF a, CF b, const Lane& L) {
  i32 k = -4 * L.kf + (L.lane == 8 ? (4 << 24) : 0);
  return cnorm(CF{a.v - b.v + k}, L);
}
template <int K>
DEV CF cshl(CF a, const Lane& L) { return cnorm(CF{a.v << K}, L); }      // a * 2^K, K <= 1 on N values

// Jacobian doubling for a = 0, general Z (dbl-2009-l, the operation count of short.js:668-737): 2M + 5S
struct CJ { CF X, Y, Z; };
DEV CJ cdbl(CJ p, const Lane& L) {
  CF a = csqr(p.X, L);
  CF b = csqr(p.Y, L);
  CF c = csqr(b, L);
  CF xb = cnorm(cadd(p.X, b), L);
  CF d = csub(csub(csqr(xb, L), a, L), c, L);
  d = cshl<1>(d, L);
  CF e = cnorm(CF{a.v * 3}, L);
  CF f = csqr(e, L);
  CF x3 = csub(f, cshl<1>(d, L), L);
  CF c8 = cshl<1>(cshl<1>(cshl<1>(c, L), L), L);
  CF y3 = csub(cmul(e, csub(d, x3, L), L), c8, L);
  CF z3 = cshl<1>(cmul(p.Y, p.Z, L), L);
  return CJ{x3, y3, z3};
}

// kind 0 mul chain, 1 sqr chain, 2 add + sub (normalised) chain, 3 doubling chain.  One wave per
// block; blocks = 1 is the lone wave of a single call.  out: the limbs of the chain's value.
template <int KIND>
__global__ void __launch_bounds__(64) k_chain(i32* out, int iters, u32 seed) {
  const int lane = threadIdx.x & 15;
  const Lane L = make_lane(lane);
  // every 16-lane row of the wave carries the same item (the DPP rows are independent)
  auto limb = [&](u32 s, int l) -> i32 {
    u32 x = s * 2654435761u + (u32)l * 0x9E3779B9u;
    x ^= x >> 15; x *= 0x85EBCA6Bu; x ^= x >> 13;
    return l > 8 ? 0 : (i32)(x & (l == 8 ? 0xFFFFFFu : (u32)M29));
  };
  CF x{limb(seed, lane)}, y{limb(seed + 1, lane)}, z{limb(seed + 2, lane)};
  CJ p{x, y, z};
#pragma nounroll
  for (int it = 0; it < iters; it++) {
    if (KIND == 0) x = cmul(x, y, L);
    else if (KIND == 1) x = csqr(x, L);
    else if (KIND == 2) { x = cnorm(cadd(x, y), L); x = csub(x, z, L); }
    else p = cdbl(p, L);
  }
  if (threadIdx.x < 16) {
    i32* o = out + (size_t)blockIdx.x * 64;
    o[threadIdx.x] = KIND == 3 ? p.X.v : x.v;
    o[16 + threadIdx.x] = p.Y.v;
    o[32 + threadIdx.x] = p.Z.v;
    o[48 + threadIdx.x] = y.v;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int KIND>
static double run(int blocks, int iters, u32 seed, std::vector<i32>* res) {
  i32* d;
  CK(hipMalloc(&d, (size_t)blocks * 64 * 4));
  CK(hipMemset(d, 0, (size_t)blocks * 64 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_chain<KIND>, dim3(blocks), dim3(64), 0, 0, d, iters, seed);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (rep && ms < best) best = ms;
  }
  if (res) { res->resize(64); CK(hipMemcpy(res->data(), d, 64 * 4, hipMemcpyDeviceToHost)); }
  CK(hipFree(d));
  return (double)best * 1e6 / iters;                  // ns per chained operation
}

static void dump(const char* name, const std::vector<i32>& r, int off, bool last) {
  printf("\"%s\": [", name);
  for (int i = 0; i < 9; i++) printf("%d%s", r[off + i], i == 8 ? "" : ", ");
  printf("]%s", last ? "" : ", ");
}

int main(int argc, char** argv) {
  const u32 seed = argc > 1 ? (u32)atoi(argv[1]) : 12345u;
  std::vector<i32> r;
  printf("{\"seed\": %u, \"checks\": {", seed);
  // results of SHORT chains for the driver's exact check (value = sum limb * 2^(29 l) mod p)
  for (int n = 1; n <= 3; n++) {
    run<0>(1, n, seed, &r); printf("\"mul%d\": {", n); dump("x", r, 0, false); dump("y", r, 48, true); printf("}, ");
    run<1>(1, n, seed, &r); printf("\"sqr%d\": {", n); dump("x", r, 0, true); printf("}, ");
    run<2>(1, n, seed, &r); printf("\"addsub%d\": {", n); dump("x", r, 0, true); printf("}, ");
    run<3>(1, n, seed, &r); printf("\"dbl%d\": {", n); dump("X", r, 0, false); dump("Y", r, 16, false); dump("Z", r, 32, true); printf("}%s", n == 3 ? "" : ", ");
  }
  printf("}, \"ns_per_op\": {");
  const int blocks[4] = {1, 256, 1024, 4096};
  for (int b = 0; b < 4; b++) {
    printf("\"%d\": {\"mul\": %.2f, \"sqr\": %.2f, \"addsub\": %.2f, \"dbl\": %.2f}%s", blocks[b],
           run<0>(blocks[b], 20000, seed, nullptr), run<1>(blocks[b], 20000, seed, nullptr),
           run<2>(blocks[b], 20000, seed, nullptr), run<3>(blocks[b], 3000, seed, nullptr), b == 3 ? "" : ", ");
  }
  printf("}}\n");
  return 0;
}
