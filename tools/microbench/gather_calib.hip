// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS engine's access pattern (the
// guide's x2 correction of FETCH_SIZE is measured for wide coalesced streams only): every lane
// gathers one 64-byte table entry (four 16-byte loads, as the ladders' `A q = tbl[idx]` does) from
// a 2 GiB table -- larger than the 256 MiB Infinity Cache -- at a pseudo-random 64-byte-aligned
// index, and writes one 64-byte entry at a 1 KiB stride (the per-lane window tables).  The bytes
// are known exactly: lanes x 64 each way.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- gather_calib      (and WRITE_SIZE)
// tools/refresh_profiles.py divides the known bytes by the counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct E { uint32_t w[16]; };

__global__ void k_gather64(const E* tbl, uint64_t entries, uint32_t* out, uint32_t seed) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint64_t idx = ((i * 0x9E3779B97F4A7C15ull + seed) >> 17) % entries;
  E e = tbl[idx];
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) x ^= e.w[j];
  out[i] = x;
}
__global__ void k_scatter64(E* tbl, uint32_t seed) {           // lane i writes entry 16*i (1 KiB stride)
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  E e;
#pragma unroll
  for (int j = 0; j < 16; j++) e.w[j] = seed + (uint32_t)i * 16 + j;
  tbl[i * 16] = e;
}

int main() {
  const uint64_t entries = (2ull << 30) / sizeof(E);             // 2 GiB
  const uint64_t lanes = 1ull << 21;                             // x 1 KiB stride = the whole 2 GiB for the scatter
  E* tbl; uint32_t* out;
  if (hipMalloc(&tbl, entries * sizeof(E)) != hipSuccess || hipMalloc(&out, lanes * 4 * 2) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(tbl, 1, entries * sizeof(E));
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(k_gather64, dim3((lanes * 2) / 128), dim3(128), 0, 0, tbl, entries, out, 12345u + rep);
    hipLaunchKernelGGL(k_scatter64, dim3(lanes / 128), dim3(128), 0, 0, tbl, 777u + rep);
  }
  hipDeviceSynchronize();
  printf("{\"gather_lanes\": %llu, \"gather_bytes_per_dispatch\": %llu, \"scatter_lanes\": %llu, \"scatter_bytes_per_dispatch\": %llu, \"out_bytes_per_gather_dispatch\": %llu}\n",
         (unsigned long long)(lanes * 2), (unsigned long long)(lanes * 2 * 64), (unsigned long long)lanes,
         (unsigned long long)(lanes * 64), (unsigned long long)(lanes * 2 * 4));
  return 0;
}
