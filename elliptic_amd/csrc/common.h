// ellgpu -- common definitions shared by every device header.
//
// All arithmetic code in csrc/*.h is written as plain C++ templates marked
// ELL_HD.  hipcc compiles them for gfx950 (the product: libellgpu.so); the
// CPU-only unit tests additionally compile the very same headers with g++
// (tests/hostsim/) to check the kernels' logic item-by-item against the
// oracle without a GPU.  That host build is test infrastructure -- it is not
// linked into libellgpu.so, which has no CPU fallback.
#pragma once
// Register-pressure switches of the secp256k1 ladder kernels (engine.h: ELL_ENDO_MIN_WAVES):
//   ELL_BETA_REMAT   beta moved in from scalar registers at every lambda*P lookup instead of
//                    eight VGPRs held across the ladder
//   ELL_SPILL_ZG     the table's common Z parked in a free table slot during the ladder
//   ELL_LATE_LOADS   u1 / k1 and r loaded after the ladder, behind compiler barriers
// FpSolinas::reduce_wide (experiment, off): product words join the signed 64-bit accumulators
// through v_mad_u64_u32 (w * 1 + acc) and carries come from v_ashrrev_i64, instead of widening
// every product word to a register pair first.  4.5 % fewer VALU instructions in the p384 ladder
// kernel -- and 4-8 % SLOWER on all four NIST curves (profiles/r02_solinas_mad_fold_ab.txt): the
// extra 2 000 multiplier-pipe instructions cost more than the 2 600 moves they replace.
#ifndef ELL_SOLINAS_MAD_FOLD
#define ELL_SOLINAS_MAD_FOLD 0
#endif
// FpSolinas::reduce_wide of p192 / p224 / p384: 1 = the two-stage carry-chain fold
// (reduce_wide_chain), 0 = the lazy-accumulator fold for every Solinas prime (p256 always)
#ifndef ELL_SOLINAS_CHAIN
#define ELL_SOLINAS_CHAIN 1
#endif
// Ladder::build_table_odd8: 1 = the odd multiples are chained with co-Z additions (4M + 2S each),
// 0 = with mixed additions (8M + 3S)
// window width of the secp256k1 odd-digit ladder (Work::Endo): 4 = 33 windows over 8 odd
// multiples (the full-grid tuning), 5 = 26 windows over 16 (the small-grid tuning, WIDE)
#ifndef ELL_ENDO_WBITS
#define ELL_ENDO_WBITS 4
#endif
#ifndef ELL_ENDO_WBITS_WIDE
#define ELL_ENDO_WBITS_WIDE 5
#endif
#ifndef ELL_P384_WBITS
#define ELL_P384_WBITS 4            // window width of the p384 ladder (experiment: 5)
#endif
#ifndef ELL_COZ_TABLE
#define ELL_COZ_TABLE 1
#endif
#ifndef ELL_P224_TS_WINDOW
#define ELL_P224_TS_WINDOW 1      // p224 square root: windowed Tonelli-Shanks (fp.h)
#endif
#ifndef ELL_BETA_REMAT
#define ELL_BETA_REMAT 1
#endif
#ifndef ELL_SPILL_ZG
#define ELL_SPILL_ZG 1
#endif
#ifndef ELL_LATE_LOADS
#define ELL_LATE_LOADS 1
#endif
// The opposite trade is the small-grid tuning (template parameter WIDE of the secp256k1 verify /
// P*k kernels, engine.h): more registers, beta / zg / u1 resident, table and comb entries requested
// one step ahead of their use (Ladder::run_odd_w4, comb_add).
//   ELL_PREFETCH     developer switch: 1 forces the one-step-ahead requests in EVERY ladder
//                    kernel of a build (how the p384 A/B of DESIGN.md section 9 was made)
#ifndef ELL_PREFETCH
#define ELL_PREFETCH 0
#endif

#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ELL_HD __host__ __device__ __forceinline__
#define ELL_HD_NOINLINE __host__ __device__ __noinline__
#define ELL_UNROLL _Pragma("unroll")
#define ELL_NOUNROLL _Pragma("nounroll")
#else
#define ELL_HD inline
#define ELL_HD_NOINLINE __attribute__((noinline))
#define ELL_UNROLL
#define ELL_NOUNROLL
#endif

#define ELL_UNLIKELY(x) __builtin_expect(!!(x), 0)

namespace ell {

// Lanes past the end of a range of n items: on the device a lane whose WAVE still holds an item
// of the range does not sit the kernel out, it joins the range's last item (same loads, same
// arithmetic, same stores of the same values, in lockstep with that item's own lane -- lanes of
// one wave cannot overtake each other).  Measured on gfx950 (profiles/r04_lane_fill_ab.jsonl): a
// wave with at most 32 active lanes runs this instruction mix 16 % SLOWER than a full one --
// one verify 0.80 ms, sixty-four 0.70 ms -- so the last wave of every range is kept full.  Waves
// entirely past the end exit (a second wave redoing an item could interleave with the first one's
// reuse of its scratch slots).  Ranges start at multiples of 64 lanes.  Returns whether the lane works.
ELL_HD bool fill_lane(size_t& i, size_t n) {
  if (i < n) return true;
#if defined(__HIP_DEVICE_COMPILE__)
  if ((i & ~(size_t)63) >= n) return false;
  i = n - 1;
  return true;
#else
  return false;
#endif
}

typedef uint32_t u32;
typedef uint64_t u64;
typedef uint8_t u8;

// ---- multi-limb primitives (little-endian 32-bit limbs) -------------------

// Carry primitives.  clang's __builtin_addc/__builtin_subc lower to genuine
// v_add_co_u32 / v_addc_co_u32 chains on gfx950 (the u64-widening idiom does
// not: it becomes v_lshl_add_u64 + v_mov pairs, 4x the instructions) and the
// compiler pads the VCC read-after-write hazard itself.
ELL_HD u32 addc32(u32 a, u32 b, u32 cin, u32& cout) {
#if defined(__clang__)
  unsigned co;
  u32 r = __builtin_addc(a, b, cin, &co);
  cout = co;
  return r;
#else
  u64 t = (u64)a + b + cin;
  cout = (u32)(t >> 32);
  return (u32)t;
#endif
}
ELL_HD u32 subb32(u32 a, u32 b, u32 bin, u32& bout) {
#if defined(__clang__)
  unsigned bo;
  u32 r = __builtin_subc(a, b, bin, &bo);
  bout = bo;
  return r;
#else
  u64 t = (u64)a - b - bin;
  bout = (u32)(t >> 63);
  return (u32)t;
#endif
}

// r = a + b, returns carry out
template <int L>
ELL_HD u32 bn_add(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L]) {
  u32 c = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) r[i] = addc32(a[i], b[i], c, c);
  return c;
}

// r = a - b, returns borrow out (0/1)
template <int L>
ELL_HD u32 bn_sub(u32 (&r)[L], const u32 (&a)[L], const u32 (&b)[L]) {
  u32 br = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) r[i] = subb32(a[i], b[i], br, br);
  return br;
}

// a >= b ?
template <int L>
ELL_HD bool bn_geq(const u32 (&a)[L], const u32 (&b)[L]) {
  u32 br = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) (void)subb32(a[i], b[i], br, br);
  return br == 0;
}

template <int L>
ELL_HD bool bn_is_zero(const u32 (&a)[L]) {
  u32 o = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) o |= a[i];
  return o == 0;
}

template <int L>
ELL_HD bool bn_eq(const u32 (&a)[L], const u32 (&b)[L]) {
  u32 o = 0;
  ELL_UNROLL
  for (int i = 0; i < L; i++) o |= a[i] ^ b[i];
  return o == 0;
}

template <int L>
ELL_HD void bn_copy(u32 (&r)[L], const u32 (&a)[L]) {
  ELL_UNROLL
  for (int i = 0; i < L; i++) r[i] = a[i];
}

template <int L>
ELL_HD void bn_zero(u32 (&r)[L]) {
  ELL_UNROLL
  for (int i = 0; i < L; i++) r[i] = 0;
}

// r = c ? a : b   (branch-free; compiles to v_cndmask)
template <int L>
ELL_HD void bn_select(u32 (&r)[L], bool c, const u32 (&a)[L], const u32 (&b)[L]) {
  ELL_UNROLL
  for (int i = 0; i < L; i++) r[i] = c ? a[i] : b[i];
}

// full product r[0..LA+LB) = a * b   (operand scanning; one v_mad_u64_u32 per
// partial product, the 64-bit accumulator carries the row's running carry)
template <int LA, int LB>
ELL_HD void bn_mul_wide(u32 (&r)[LA + LB], const u32 (&a)[LA], const u32 (&b)[LB]) {
  {
    u32 carry = 0;
    ELL_UNROLL
    for (int j = 0; j < LB; j++) {
      u64 t = (u64)a[0] * b[j] + carry;
      r[j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    r[LB] = carry;
  }
  ELL_UNROLL
  for (int i = 1; i < LA; i++) {
    u32 carry = 0;
    ELL_UNROLL
    for (int j = 0; j < LB; j++) {
      u64 t = (u64)a[i] * b[j] + r[i + j] + carry;
      r[i + j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    r[i + LB] = carry;
  }
}

// full square r[0..2L) = a^2: off-diagonal products once, doubled, plus the
// diagonal (L(L-1)/2 + L multiplies instead of L^2)
template <int L>
ELL_HD void bn_sqr_wide(u32 (&r)[2 * L], const u32 (&a)[L]) {
  ELL_UNROLL
  for (int i = 0; i < 2 * L; i++) r[i] = 0;
  // off-diagonal: sum_{i<j} a_i a_j 2^(32(i+j))
  ELL_UNROLL
  for (int i = 0; i < L - 1; i++) {
    u32 carry = 0;
    ELL_UNROLL
    for (int j = i + 1; j < L; j++) {
      u64 t = (u64)a[i] * a[j] + r[i + j] + carry;
      r[i + j] = (u32)t;
      carry = (u32)(t >> 32);
    }
    r[i + L] = carry;
  }
  // double
  {
    u32 top = 0;
    ELL_UNROLL
    for (int i = 1; i < 2 * L; i++) {
      u32 v = r[i];
      r[i] = (v << 1) | top;
      top = v >> 31;
    }
  }
  // add diagonal squares
  {
    u32 carry = 0;
    ELL_UNROLL
    for (int i = 0; i < L; i++) {
      u64 t = (u64)a[i] * a[i] + r[2 * i] + carry;
      r[2 * i] = (u32)t;
      u64 u = (u64)r[2 * i + 1] + (u32)(t >> 32);
      r[2 * i + 1] = (u32)u;
      carry = (u32)(u >> 32);
    }
  }
}

// ---- byte <-> limb conversion (big-endian bytes at the API) ---------------

ELL_HD u32 bswap32(u32 v) { return __builtin_bswap32(v); }

// nb bytes big-endian -> L limbs little-endian (nb <= 4L; high limbs zero).
// When nb is a multiple of 4 and p is 4-byte aligned (every fixed-width field
// of the C ABI except the 66-byte p521 ones) this is L dword loads + bswaps.
template <int L>
ELL_HD void load_be(u32 (&r)[L], const u8* p, int nb) {
  if ((nb & 3) == 0 && (((uintptr_t)p) & 3) == 0) {
    const u32* w = (const u32*)p;
    int nw = nb >> 2;
    ELL_UNROLL
    for (int i = 0; i < L; i++) r[i] = (i < nw) ? bswap32(w[nw - 1 - i]) : 0u;
    return;
  }
  ELL_UNROLL
  for (int i = 0; i < L; i++) {
    u32 v = 0;
    ELL_UNROLL
    for (int b = 0; b < 4; b++) {
      int idx = nb - 1 - (4 * i + b);          // byte of weight 2^(8(4i+b))
      if (idx >= 0) v |= (u32)p[idx] << (8 * b);
    }
    r[i] = v;
  }
}

template <int L>
ELL_HD void store_be(u8* p, const u32 (&a)[L], int nb) {
  if ((nb & 3) == 0 && (((uintptr_t)p) & 3) == 0) {
    u32* w = (u32*)p;
    int nw = nb >> 2;
    ELL_UNROLL
    for (int i = 0; i < L; i++)
      if (i < nw) w[nw - 1 - i] = bswap32(a[i]);
    return;
  }
  ELL_UNROLL
  for (int i = 0; i < L; i++) {
    ELL_UNROLL
    for (int b = 0; b < 4; b++) {
      int idx = nb - 1 - (4 * i + b);
      if (idx >= 0) p[idx] = (u8)(a[i] >> (8 * b));
    }
  }
}

// little-endian bytes (ed25519 / x25519 wire order is handled by the host
// layer; the C ABI is big-endian everywhere, see include/ellgpu.h)

}  // namespace ell
