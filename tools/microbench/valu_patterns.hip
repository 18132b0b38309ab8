// Micro-benchmark of the instruction patterns inside the field multiply (gfx950): how many
// SIMD cycles per instruction for v_mad_u64_u32 / v_addc_co_u32 in the combinations the
// generated blocks use.  One wavefront per block; blocks = 1024 * w puts w waves on each SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o valu_patterns valu_patterns.hip && ./valu_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32;
typedef uint64_t u64;

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))

template <int P>
__global__ void __launch_bounds__(64) k(u32* out, int iters, u32 seed) {
  u32 a = seed + threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u;
  u64 A0 = a, A1 = b, A2 = a + 1, A3 = b + 1, A4 = a + 2, A5 = b + 2, A6 = a + 3, A7 = b + 3;
  u32 e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
  u32 sb = seed | 1;
  for (int it = 0; it < iters; it++) {
    if (P == 1) {          // 8 independent mads, VGPR x VGPR, carry-out to one dummy SGPR pair
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n"
          "v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n"
          "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n"
          "v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "s20", "s21");
    } else if (P == 2) {   // same, one multiplicand in an SGPR
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n"
          "v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n"
          "v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n"
          "v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "s"(sb) : "s20", "s21");
    } else if (P == 3) {   // 8 independent plain adds
      asm volatile(REP4(
          "v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n"
          "v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a));
    } else if (P == 4) {   // 8 independent e64 add-with-carry, carry-in from SGPR pairs written long ago
      asm volatile(REP4(
          "v_addc_co_u32_e64 %0, s[20:21], 0, %0, s[22:23]\n v_addc_co_u32_e64 %1, s[20:21], 0, %1, s[24:25]\n"
          "v_addc_co_u32_e64 %2, s[20:21], 0, %2, s[22:23]\n v_addc_co_u32_e64 %3, s[20:21], 0, %3, s[24:25]\n"
          "v_addc_co_u32_e64 %4, s[20:21], 0, %4, s[22:23]\n v_addc_co_u32_e64 %5, s[20:21], 0, %5, s[24:25]\n"
          "v_addc_co_u32_e64 %6, s[20:21], 0, %6, s[22:23]\n v_addc_co_u32_e64 %7, s[20:21], 0, %7, s[24:25]\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : : "s20", "s21", "s22", "s23", "s24", "s25");
    } else if (P == 5) {   // the product inner pattern: mad (carry -> sK) + addc consuming it 3 slots later
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n"
          "v_mad_u64_u32 %1, s[22:23], %8, %9, %1\n"
          "v_mad_u64_u32 %2, s[24:25], %8, %9, %2\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[20:21]\n"
          "v_mad_u64_u32 %3, s[26:27], %8, %9, %3\n"
          "v_addc_co_u32_e64 %5, s[28:29], 0, %5, s[22:23]\n"
          "v_mad_u64_u32 %0, s[20:21], %9, %8, %0\n"
          "v_addc_co_u32_e64 %6, s[28:29], 0, %6, s[24:25]\n"
          "v_mad_u64_u32 %1, s[22:23], %9, %8, %1\n"
          "v_addc_co_u32_e64 %7, s[28:29], 0, %7, s[26:27]\n"
          "v_mad_u64_u32 %2, s[24:25], %9, %8, %2\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[20:21]\n"
          "v_mad_u64_u32 %3, s[26:27], %9, %8, %3\n"
          "v_addc_co_u32_e64 %5, s[28:29], 0, %5, s[22:23]\n"
          "s_nop 0\n"
          "v_addc_co_u32_e64 %6, s[28:29], 0, %6, s[24:25]\n"
          "s_nop 0\n"
          "v_addc_co_u32_e64 %7, s[28:29], 0, %7, s[26:27]\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");
    } else if (P == 6) {   // dependent VCC carry chain of 8 with the mandatory wait states
      asm volatile(REP4(
          "v_add_co_u32 %0, vcc, %8, %0\n s_nop 1\n v_addc_co_u32 %1, vcc, %8, %1, vcc\n s_nop 1\n"
          "v_addc_co_u32 %2, vcc, %8, %2, vcc\n s_nop 1\n v_addc_co_u32 %3, vcc, %8, %3, vcc\n s_nop 1\n"
          "v_addc_co_u32 %4, vcc, %8, %4, vcc\n s_nop 1\n v_addc_co_u32 %5, vcc, %8, %5, vcc\n s_nop 1\n"
          "v_addc_co_u32 %6, vcc, %8, %6, vcc\n s_nop 1\n v_addc_co_u32 %7, vcc, %8, %7, vcc\n s_nop 1\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a) : "vcc");
    } else if (P == 8) {   // single accumulator: every mad depends on the previous one, addc two slots behind
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[22:23]\n"
          "v_mad_u64_u32 %0, s[22:23], %9, %8, %0\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[20:21]\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[22:23]\n"
          "v_mad_u64_u32 %0, s[22:23], %9, %8, %0\n"
          "v_addc_co_u32_e64 %4, s[28:29], 0, %4, s[20:21]\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "s28", "s29");
    } else if (P == 9) {   // 8 independent moves
      asm volatile(REP4(
          "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n"
          "v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a));
    } else if (P == 10) {  // 8 independent funnel shifts
      asm volatile(REP4(
          "v_alignbit_b32 %0, %8, %0, 3\n v_alignbit_b32 %1, %8, %1, 3\n v_alignbit_b32 %2, %8, %2, 3\n"
          "v_alignbit_b32 %3, %8, %3, 3\n v_alignbit_b32 %4, %8, %4, 3\n v_alignbit_b32 %5, %8, %5, 3\n"
          "v_alignbit_b32 %6, %8, %6, 3\n v_alignbit_b32 %7, %8, %7, 3\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a));
    } else if (P == 11) {  // 8 independent selects on an SGPR-pair mask
      asm volatile(REP4(
          "v_cndmask_b32_e64 %0, %0, %8, s[22:23]\n v_cndmask_b32_e64 %1, %1, %8, s[22:23]\n"
          "v_cndmask_b32_e64 %2, %2, %8, s[22:23]\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n"
          "v_cndmask_b32_e64 %4, %4, %8, s[22:23]\n v_cndmask_b32_e64 %5, %5, %8, s[22:23]\n"
          "v_cndmask_b32_e64 %6, %6, %8, s[22:23]\n v_cndmask_b32_e64 %7, %7, %8, s[22:23]\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a) : "s22", "s23");
    } else if (P == 12) {  // 8 independent selects on VCC (e32 encoding)
      asm volatile(REP4(
          "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
          "v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
          "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
          "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a));
    } else if (P == 13) {  // 8 independent add-with-carry-out only (v_add_co, VCC written, never read)
      asm volatile(REP4(
          "v_add_co_u32 %0, vcc, %8, %0\n v_add_co_u32 %1, vcc, %8, %1\n v_add_co_u32 %2, vcc, %8, %2\n"
          "v_add_co_u32 %3, vcc, %8, %3\n v_add_co_u32 %4, vcc, %8, %4\n v_add_co_u32 %5, vcc, %8, %5\n"
          "v_add_co_u32 %6, vcc, %8, %6\n v_add_co_u32 %7, vcc, %8, %7\n")
          : "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3), "+v"(e4), "+v"(e5), "+v"(e6), "+v"(e7)
          : "v"(a) : "vcc");
    } else if (P == 14) {  // 8 independent 64-bit adds without carry flags
      asm volatile(REP4(
          "v_lshl_add_u64 %0, %0, 0, %1\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %2, %2, 0, %3\n"
          "v_lshl_add_u64 %3, %3, 0, %4\n v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 0, %6\n"
          "v_lshl_add_u64 %6, %6, 0, %7\n v_lshl_add_u64 %7, %7, 0, %0\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7));
    } else if (P == 15) {  // 8 independent 64-bit right shifts
      asm volatile(REP4(
          "v_lshrrev_b64 %0, 3, %0\n v_lshrrev_b64 %1, 3, %1\n v_lshrrev_b64 %2, 3, %2\n v_lshrrev_b64 %3, 3, %3\n"
          "v_lshrrev_b64 %4, 3, %4\n v_lshrrev_b64 %5, 3, %5\n v_lshrrev_b64 %6, 3, %6\n v_lshrrev_b64 %7, 3, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7));
    } else if (P == 16) {  // 8 independent SIGNED mads (v_mad_i64_i32), carry-out to a dummy SGPR pair
      asm volatile(REP4(
          "v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[20:21], %8, %9, %1\n"
          "v_mad_i64_i32 %2, s[20:21], %8, %9, %2\n v_mad_i64_i32 %3, s[20:21], %8, %9, %3\n"
          "v_mad_i64_i32 %4, s[20:21], %8, %9, %4\n v_mad_i64_i32 %5, s[20:21], %8, %9, %5\n"
          "v_mad_i64_i32 %6, s[20:21], %8, %9, %6\n v_mad_i64_i32 %7, s[20:21], %8, %9, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "s20", "s21");
    } else if (P == 17) {  // 8 independent unsigned mads, carry-out to VCC
      asm volatile(REP4(
          "v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n"
          "v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
          "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n"
          "v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "vcc");
    } else if (P == 18) {  // ONE dependent chain of unsigned mads (every mad reads the previous result)
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %0, s[20:21], %9, %8, %0\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %0, s[20:21], %9, %8, %0\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %0, s[20:21], %9, %8, %0\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %0, s[20:21], %9, %8, %0\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "s20", "s21");
    } else if (P == 19) {  // TWO interleaved dependent chains
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "s20", "s21");
    } else if (P == 20) {  // 8 independent 64-bit arithmetic right shifts
      asm volatile(REP4(
          "v_ashrrev_i64 %0, 3, %0\n v_ashrrev_i64 %1, 3, %1\n v_ashrrev_i64 %2, 3, %2\n v_ashrrev_i64 %3, 3, %3\n"
          "v_ashrrev_i64 %4, 3, %4\n v_ashrrev_i64 %5, 3, %5\n v_ashrrev_i64 %6, 3, %6\n v_ashrrev_i64 %7, 3, %7\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7));
    } else if (P == 21) {  // the unsaturated column pattern: two mad chains, a 64-bit shift and a mask per 6 mads
      asm volatile(REP4(
          "v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_i64_i32 %0, s[20:21], %8, %9, %0\n v_mad_i64_i32 %1, s[20:21], %9, %8, %1\n"
          "v_and_b32 %10, 0x1fffffff, %8\n v_ashrrev_i64 %0, 29, %0\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7), "+v"(a), "+v"(b), "+v"(e0)
          : : "s20", "s21");
    } else if (P == 22) {  // same with unsigned mads and a logical shift
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %9, %8, %1\n"
          "v_and_b32 %10, 0x1fffffff, %8\n v_lshrrev_b64 %0, 29, %0\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7), "+v"(a), "+v"(b), "+v"(e0)
          : : "s20", "s21");
    } else if (P == 23) {  // 4 mads + 4 plain adds interleaved (does a cheap op hide behind a multiply?)
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_add_u32 %10, %8, %10\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_add_u32 %11, %8, %11\n"
          "v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_add_u32 %12, %8, %12\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n v_add_u32 %13, %8, %13\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7), "+v"(a), "+v"(b), "+v"(e0), "+v"(e1), "+v"(e2), "+v"(e3)
          : : "s20", "s21");
    } else if (P == 7) {   // 8 independent mads whose 64-bit accumulators sit in the same VGPR bank pattern as a, b
      asm volatile(REP4(
          "v_mad_u64_u32 %0, s[20:21], %8, %9, 0\n v_mad_u64_u32 %1, s[20:21], %8, %9, 0\n"
          "v_mad_u64_u32 %2, s[20:21], %8, %9, 0\n v_mad_u64_u32 %3, s[20:21], %8, %9, 0\n"
          "v_mad_u64_u32 %4, s[20:21], %8, %9, 0\n v_mad_u64_u32 %5, s[20:21], %8, %9, 0\n"
          "v_mad_u64_u32 %6, s[20:21], %8, %9, 0\n v_mad_u64_u32 %7, s[20:21], %8, %9, 0\n")
          : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3), "+v"(A4), "+v"(A5), "+v"(A6), "+v"(A7)
          : "v"(a), "v"(b) : "s20", "s21");
    }
  }
  u64 s = A0 ^ A1 ^ A2 ^ A3 ^ A4 ^ A5 ^ A6 ^ A7;
  out[blockIdx.x * 64 + threadIdx.x] = (u32)s ^ (u32)(s >> 32) ^ e0 ^ e1 ^ e2 ^ e3 ^ e4 ^ e5 ^ e6 ^ e7;
}

template <int P>
static double run(int waves, int iters, u32* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<P>, dim3(1024 * waves), dim3(64), 0, 0, out, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  u32* out;
  hipMalloc(&out, 1024 * 16 * 64 * 4);
  const int iters = 20000;
  const int ninstr[24] = {0, 32, 32, 32, 32, 64, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32, 32};   // VALU instructions per loop iteration
  const char* name[24] = {"", "mad vgpr*vgpr+acc", "mad vgpr*sgpr+acc", "v_add_u32", "v_addc e64 (sgpr carry)",
                          "product pattern (mad+addc)", "vcc chain + s_nop 1", "mad vgpr*vgpr+0",
                          "single-chain mad+addc", "v_mov_b32", "v_alignbit_b32", "v_cndmask e64 (sgpr mask)",
                          "v_cndmask e32 (vcc)", "v_add_co (carry out only)", "v_lshl_add_u64", "v_lshrrev_b64",
                          "v_mad_i64_i32 (signed)", "mad, carry-out to vcc", "ONE dependent mad chain", "TWO dependent mad chains",
                          "v_ashrrev_i64", "unsat column (6 signed mads + and + ashr64)", "unsat column (6 unsigned mads + and + lshr64)",
                          "4 mads + 4 v_add_u32 interleaved"};
  int dev_clock_khz = 0;
  hipDeviceGetAttribute(&dev_clock_khz, hipDeviceAttributeClockRate, 0);
  printf("clock attribute %d kHz\n", dev_clock_khz);
  for (int p = 1; p <= 23; p++) {
    printf("%-30s ns per wave-instruction per SIMD at 1,2,3,4,8 waves:", name[p]);
    for (int w : {1, 2, 3, 4, 8}) {
      double ms = 0;
      switch (p) {
        case 1: ms = run<1>(w, iters, out); break;
        case 2: ms = run<2>(w, iters, out); break;
        case 3: ms = run<3>(w, iters, out); break;
        case 4: ms = run<4>(w, iters, out); break;
        case 5: ms = run<5>(w, iters, out); break;
        case 6: ms = run<6>(w, iters, out); break;
        case 7: ms = run<7>(w, iters, out); break;
        case 8: ms = run<8>(w, iters, out); break;
        case 9: ms = run<9>(w, iters, out); break;
        case 10: ms = run<10>(w, iters, out); break;
        case 11: ms = run<11>(w, iters, out); break;
        case 12: ms = run<12>(w, iters, out); break;
        case 13: ms = run<13>(w, iters, out); break;
        case 14: ms = run<14>(w, iters, out); break;
        case 15: ms = run<15>(w, iters, out); break;
        case 16: ms = run<16>(w, iters, out); break;
        case 17: ms = run<17>(w, iters, out); break;
        case 18: ms = run<18>(w, iters, out); break;
        case 19: ms = run<19>(w, iters, out); break;
        case 20: ms = run<20>(w, iters, out); break;
        case 21: ms = run<21>(w, iters, out); break;
        case 22: ms = run<22>(w, iters, out); break;
        case 23: ms = run<23>(w, iters, out); break;
      }
      double per = ms * 1e6 / ((double)iters * ninstr[p] * w);   // ns of SIMD time per instruction
      printf(" %.3f", per);
    }
    printf("\n");
  }
  return 0;
}
