// Micro-benchmark: dependent chains of field multiplications / squarings for the saturated 8 x 32
// secp256k1 field of csrc/fp.h and for the 9 x 29-bit signed prototype (u29_proto.h), one wave per
// block, 1024 * w blocks = w waves per SIMD.  Prints SIMD time per operation in units of one
// v_mad_u64_u32 issue (measured in the same run).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I elliptic_amd/csrc -I include -o u29_probe u29_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curves.h"
#include "u29_proto.h"
#include "u29_asm.h"
using namespace ell;

template <int KIND>
__global__ void __launch_bounds__(64) k(u32* out, int iters, u32 seed) {
  u32 acc = 0;
  if (KIND < 10) {
    typedef FpK256 F;
    F::El x, y;
#pragma unroll
    for (int i = 0; i < 8; i++) { x.v[i] = seed * (i + 1) + threadIdx.x * 2654435761u; y.v[i] = (seed ^ 0x9E3779B9u) * (i + 3) + threadIdx.x; }
    x.v[7] &= 0x7FFFFFFFu; y.v[7] &= 0x7FFFFFFFu;
#pragma nounroll
    for (int it = 0; it < iters; it++) {
      if (KIND == 0) x = F::mul(x, y);
      else if (KIND == 1) x = F::sqr(x);
      else { x = F::add(x, y); x = F::sub(x, y); }
    }
#pragma unroll
    for (int i = 0; i < 8; i++) acc ^= x.v[i];
  } else if (KIND < 20) {
    typedef U29 F;
    F::El x, y;
#pragma unroll
    for (int i = 0; i < 9; i++) { x.v[i] = (seed * (i + 1) + threadIdx.x * 2654435761u) & 0x1FFFFFFF; y.v[i] = ((seed ^ 0x9E3779B9u) * (i + 3) + threadIdx.x) & 0x1FFFFFFF; }
#pragma nounroll
    for (int it = 0; it < iters; it++) {
      if (KIND == 10) x = F::mul(x, y);
      else if (KIND == 11) x = F::sqr(x);
      else if (KIND == 12) { x = F::add(x, y); x = F::sub(x, y); }
      else if (KIND == 13) x = F::norm(F::add(x, y));
      else if (KIND == 14) x = u29_mul_asm(x, y);
      else if (KIND == 15) x = u29_sqr_asm(x);
      else {                      // 16: self-check of the asm forms against the C++ forms
        F::El m1 = F::mul(x, y), m2 = u29_mul_asm(x, y), s1 = F::sqr(x), s2 = u29_sqr_asm(x);
#pragma unroll
        for (int i = 0; i < 9; i++) acc |= (u32)(m1.v[i] ^ m2.v[i]) | (u32)(s1.v[i] ^ s2.v[i]);
        x = F::sub(m1, s1); x = F::norm(x);
      }
    }
    if (KIND == 16) { out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc; return; }
#pragma unroll
    for (int i = 0; i < 9; i++) acc ^= x.v[i];
  } else {
    u32 a = seed + threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u;
    u64 A[16];
#pragma unroll
    for (int j = 0; j < 16; j++) A[j] = (u64)(a + j) << 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int j = 0; j < 16; j++) A[j] = (u64)a * b + A[j];
      a += 0x1234567u; b ^= a;
    }
#pragma unroll
    for (int j = 0; j < 16; j++) acc ^= (u32)A[j] ^ (u32)(A[j] >> 32);
  }
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = acc;
}

template <int KIND>
static double run(int waves, int iters, u32* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(1024 * waves), dim3(64), 0, 0, out, iters, 12345u);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  u32* out; hipMalloc(&out, 1024 * 16 * 64 * 4);
  // mad issue time: 16 independent mads per iteration, 8 waves / SIMD
  double ms = run<20>(8, 4096, out);
  double mad_ns = ms * 1e6 / (4096.0 * 16 * 8);        // ns of SIMD time per wave-mad
  printf("v_mad_u64_u32 issue: %.3f ns per wave-instruction per SIMD (%.2f T mad/s)\n", mad_ns, 1024 * 64 / mad_ns / 1e3);
  const char* names[] = {"sat 8x32 mul", "sat 8x32 sqr", "sat 8x32 add+sub (per op)", "u29 mul", "u29 sqr", "u29 add+sub (per op)", "u29 add+norm (per pair)", "u29 mul asm", "u29 sqr asm"};
  int it = 3000;
  {   // asm forms == C++ forms on 64 K lanes x 200 chained operand pairs (signed limbs included)
    run<16>(1, 200, out);
    static u32 host[1024 * 64];
    hipMemcpy(host, out, sizeof host, hipMemcpyDeviceToHost);
    u32 bad = 0;
    for (size_t i = 0; i < 1024 * 64; i++) bad |= host[i];
    printf("asm vs C++ self-check: %s\n", bad ? "MISMATCH" : "identical");
  }
  for (int w : {2, 3, 4, 8}) {
    double t[9];
    t[0] = run<0>(w, it, out); t[1] = run<1>(w, it, out); t[2] = run<2>(w, it, out) / 2;
    t[3] = run<10>(w, it, out); t[4] = run<11>(w, it, out); t[5] = run<12>(w, it, out) / 2; t[6] = run<13>(w, it, out); t[7] = run<14>(w, it, out); t[8] = run<15>(w, it, out);
    printf("waves/SIMD %d:", w);
    for (int i = 0; i < 9; i++) printf("  %s %.1f", names[i], t[i] * 1e6 / ((double)it * w) / mad_ns);
    printf("\n");
  }
  return 0;
}
