// ellgpu -- host-side engine: batch orchestration above the per-item work
// functions.  Templated on a backend BK that supplies memory and launches:
//
//   HipBackend  (capi.hip)          hipMalloc / hipMemcpyAsync / kernel launches on
//                                   one stream of one MI355X -- the product.
//   LoopBackend (tests/hostsim)     malloc / memcpy / a for-loop over thread ids --
//                                   CPU-only unit tests of the same code; never
//                                   shipped, never selected at run time.
//
// A launch is `bk.launch(functor, nthreads)`: the functor's operator()(tid, ds)
// runs once per thread id with a DigitStore of Fn::DS_PER_LANE bytes per lane.
#pragma once

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "edwards.h"
#include "edcustom.h"
#include "mont.h"
#include "work.h"
#include "coop_ed.h"

#ifndef ELL_INV_BATCH
#define ELL_INV_BATCH 16
#endif
// p521 (one wave per SIMD, 256 VGPRs + AGPRs, spill-free) issues at the lone-wave half rate.  A
// second instantiation held to 256 registers runs two waves per SIMD: each wave is 1.6x slower
// (spills), the pair 1.26x faster -- but only batches of more than one full single-wave round
// (256 CUs x 4 SIMDs x 64 lanes) have a second wave to pair.
#define ELL_P521_PAIR_MIN ((size_t)256 * 4 * 64 + 4096)
#ifndef ELL_MULVAR_MIN_WAVES
#define ELL_MULVAR_MIN_WAVES 4
#endif
#ifndef ELL_CUSTOM_MIN_WAVES
#define ELL_CUSTOM_MIN_WAVES 3      // user-defined curves (Jacobian window table, generic-a doubling)
#endif
// secp256k1's verify / k1*G + k2*P kernels: 4 waves/SIMD (128 VGPRs).  With beta re-materialised
// at each lookup, zg parked in a free table slot and u1 / r loaded behind compiler barriers
// (common.h: ELL_BETA_REMAT, ELL_SPILL_ZG, ELL_LATE_LOADS) the ladder loop holds no spills at 128
// registers: ecdsa_main 8.43 -> 8.20 ms per 2^20 (3 waves, 168 VGPRs before; 5 waves = 96 VGPRs
// spills 304 B and is 13 % slower), profiles/r02_four_waves_ab.jsonl
#ifndef ELL_ENDO_MIN_WAVES
#define ELL_ENDO_MIN_WAVES 4
#endif
#ifndef ELL_PIPE_STEP_DEFAULT
#ifdef ELL_PIPE_STEP
#define ELL_PIPE_STEP_DEFAULT ELL_PIPE_STEP
#else
#define ELL_PIPE_STEP_DEFAULT 3      // hip_backend.h documents the sweep
#endif
#endif
#ifndef ELL_FIXED_MIN_WAVES
#define ELL_FIXED_MIN_WAVES 4       // fixed-base comb kernels (mul_fixed, sign_mul) of the <= 256-bit curves: 128 VGPRs, -2..3.5 % on secp256k1, neutral elsewhere
#endif
#ifndef ELL_ECDSA_MIN_WAVES
#define ELL_ECDSA_MIN_WAVES 3
#endif
// (batches of at most three waves on every SIMD take the WIDE ecdsa_main: Engine::Tuning)
// ecdsa_table (the window table of the small-grid verify, built beside ecdsa_prep): waves per
// SIMD its register budget leaves room for
#ifndef ELL_ECDSA_TABLE_MIN_WAVES
#define ELL_ECDSA_TABLE_MIN_WAVES 4
#endif
// 1 (default): small-grid secp256k1 verifies run as prep || table -> ladder; 0: prep -> main
#ifndef ELL_SPLIT_SMALL_VERIFY
#define ELL_SPLIT_SMALL_VERIFY 1
#endif
// entries per slice of the fixed-base table build (Engine::ensure_comb); the CPU unit-test build
// of these headers passes a small value so that its narrow combs are built in several slices too
#ifndef ELL_COMB_SLICE
#define ELL_COMB_SLICE (1u << 20)
#endif
#ifndef ELL_P521_MIN_WAVES
#define ELL_P521_MIN_WAVES 1        // (the p521 ladders take 232-234 VGPRs: two waves either way; 3 waves spill 560 B)
#endif
// p384 (12 limbs): waves per SIMD the ladder kernels leave room for
#ifndef ELL_P384_MIN_WAVES
#define ELL_P384_MIN_WAVES 3        // 168 VGPRs: +2..3 % over 2 waves since the leaner add / sub (profiles/r02_p384_waves_ab.txt)
#endif

namespace ell {

enum { E_OK = 0, E_NODEVICE = -1, E_ARG = -2, E_HIP = -3, E_NOMEM = -4, E_UNSUPPORTED = -5 };

// ---- functors (one per kernel) ------------------------------------------------
// WIDE: the small-grid tuning of the secp256k1 ladder (see FnEcdsaMain)
template <class CV, int MW = 0, bool WIDE = false>
struct FnMulVar {
  static constexpr const char* NAME = "mul_var";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = W::NWIN * W::NSV;
  static constexpr int MIN_WAVES = MW ? MW : (W::L <= 8 ? (CV::JTABLE ? ELL_CUSTOM_MIN_WAVES : ELL_MULVAR_MIN_WAVES) : (W::L == 12 ? ELL_P384_MIN_WAVES : ELL_P521_MIN_WAVES));   // <= 128 VGPRs for 256-bit curves: +2..4 % despite ~50 B of spills
  size_t n; const u8* k; const u8* xy; typename W::VT* tbl; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) W::template mul_var<WIDE>(i, n, k, xy, tbl, ds, jac);
  }
};
template <class CV>
struct FnMulAdd2 {
  static constexpr const char* NAME = "mul_add2";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = W::NWIN * W::NSV * 2;
  static constexpr int MIN_WAVES = W::L <= 8 ? 3 : (W::L == 12 ? 2 : 1);   // <= 168 VGPRs for the 256-bit curves, <= 256 for p384
  size_t n; const u8* k1; const u8* xy1; const u8* k2; const u8* xy2; typename W::J* tbl; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) W::mul_add2(i, n, k1, xy1, k2, xy2, tbl, ds, jac);
  }
};
template <class CV, int MW = 0>
struct FnMulAddG {
  static constexpr const char* NAME = "mul_add_g";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = W::NWIN * W::NSV;
  static constexpr int MIN_WAVES = MW ? MW : (W::L <= 8 ? (CV::ENDO ? ELL_ENDO_MIN_WAVES : ELL_ECDSA_MIN_WAVES) : (W::L == 12 ? ELL_P384_MIN_WAVES : ELL_P521_MIN_WAVES));      // p384: 2 waves/SIMD (<= 256 registers) beats a spill-free single wave
  size_t n; const u8* k1; const u8* k2; const u8* xy2; const typename W::A* comb;
  typename W::VT* tbl; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) W::mul_add_g_item(i, n, k1, k2, xy2, comb, tbl, ds, jac);
  }
};
template <class CV, int MW = 0>
struct FnMulFixed {
  static constexpr const char* NAME = "mul_fixed";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = MW ? MW : (W::L <= 8 ? ELL_FIXED_MIN_WAVES : 1);
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* k; const typename W::A* comb; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::mul_fixed(i, n, k, comb, jac);
  }
};
// comb construction: entry idx = w * PER + (d - 1) of the fixed-base table is (d << (CB w)) * G;
// this writes the scalar (reduced mod n) and the generator for entries [first, first + n)
template <class CV>
struct FnCombGen {
  static constexpr const char* NAME = "comb_gen";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; size_t first; u8* k; u8* xy; int cb;       // cb: window bits of a SIGNED comb (else W::COMB_BITS)
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i >= n) return;
    const size_t idx = first + i;
    const size_t per = W::COMB_SIGNED ? ((size_t)1 << (cb - 1)) : (size_t)W::COMB_DIG;
    const int w = (int)(idx / per);
    const u32 d = (u32)(idx % per) + 1u;
    const int sh = w * (W::COMB_SIGNED ? cb : W::COMB_BITS);
    // (d << sh) mod n: the value can exceed the scalar's byte length in the top window (the
    // signed recoding's carry window of a narrow comb: 2^256 * G), so it goes through the order
    // field like any over-long byte string (Work::bytes_mod_n)
    constexpr int LN = W::LN;
    u32 v[2 * LN];
    ELL_UNROLL
    for (int l = 0; l < 2 * LN; l++) v[l] = 0;
    const int li = sh >> 5, bs = sh & 31;
    ELL_UNROLL
    for (int l = 0; l < 2 * LN; l++) {
      if (l == li) v[l] = d << bs;
      if (l == li + 1 && bs) v[l] = d >> (32 - bs);
    }
    u8 vb[8 * LN];
    store_be<2 * LN>(vb, v, 8 * LN);
    typename W::Nl km = W::bytes_mod_n(vb, 8 * LN);
    u32 kn[LN];
    W::Fn::to_plain(kn, km);
    u32 kk[W::L], gx[W::L], gy[W::L];
    ELL_UNROLL
    for (int l = 0; l < W::L; l++) { kk[l] = l < LN ? kn[l] : 0u; gx[l] = W::C::gx_plain[l]; gy[l] = W::C::gy_plain[l]; }
    store_be<W::L>(k + i * W::BYTES, kk, W::BYTES);
    store_be<W::L>(xy + i * 2 * W::BYTES, gx, W::BYTES);
    store_be<W::L>(xy + i * 2 * W::BYTES + W::BYTES, gy, W::BYTES);
  }
};
template <class CV>
struct FnNormalize {
  static constexpr const char* NAME = "normalize";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u32* jac; u32* pre; u8* out_xy; u8* out_inf;
  typename W::A* raw;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) W::normalize(t, T, n, K, jac, pre, out_xy, out_inf, raw);
  }
};
// post-pass of the point-valued calls: operands that are not on the curve are outside the
// engine's domain -> out_inf = 2, result zeroed (Work::domain_mark)
template <class CV>
struct FnDomainMark {
  static constexpr const char* NAME = "domain_mark";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* xy2; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::domain_mark(i, xy1, xy2, out_xy, out_inf);
  }
};
struct FnEdDomainMark {
  static constexpr const char* NAME = "ed_domain_mark";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* xy2; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::domain_mark(i, xy1, xy2, out_xy, out_inf);
  }
};
struct FnEdcDomainMark {
  static constexpr const char* NAME = "edc_domain_mark";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* xy2; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdcWork::domain_mark(i, xy1, xy2, out_xy, out_inf);
  }
};
// MW != 0: a second instantiation held to 512 / MW registers (default scheduling strategy): the
// one that runs BESIDE ecdsa_table in the small-grid verify (FnEcdsaPrepTable) -- at 216
// registers two waves of ecdsa_prep leave a SIMD no room for anything else, and the two jobs
// would take turns instead of sharing it (profiles/r04_split_verify_ab.txt)
template <class CV, int MW = 0>
struct FnEcdsaPrep {
  static constexpr const char* NAME = "ecdsa_prep";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = MW ? MW : 1;
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u8* hash; int hash_len; int shift; const u8* r; const u8* s;
  u32* pre; u32* u12; u8* valid;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) W::ecdsa_prep(t, T, n, K, hash, hash_len, shift, r, s, pre, u12, valid);
  }
};
// WIDE: the small-grid tuning (secp256k1 only; Engine::ecdsa_chunk picks it for batches of at most
// ELL_SMALL_GRID items): 3 waves/SIMD worth of registers, beta / zg / u1 resident, table and comb
// entries requested one step ahead.  Same results; 3 % faster where only two waves per SIMD are
// resident and gather latency is exposed, 0.6 % slower on a full grid (profiles/r03_small_grid_ab.jsonl).
template <class CV, int MW = 0, bool WIDE = false>
struct FnEcdsaMain {
  static constexpr const char* NAME = "ecdsa_main";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = MW ? MW : (W::L <= 8 ? (CV::ENDO ? ELL_ENDO_MIN_WAVES : ELL_ECDSA_MIN_WAVES) : (W::L == 12 ? ELL_P384_MIN_WAVES : ELL_P521_MIN_WAVES));   // 128 VGPRs for secp256k1, <= 168 for the other 256-bit curves, <= 256 for p384
  static constexpr int DS_PER_LANE = W::NWIN * W::NSV;
  size_t n; const u32* u12; const u8* valid; const u8* r; const u8* pub;
  const typename W::A* comb; typename W::VT* tbl; u8* ok; u8* st;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) W::template ecdsa_main<WIDE>(i, n, u12, valid, r, pub, comb, tbl, ds, ok, st);
  }
};
// the small-grid verify in two kernels (Work::ecdsa_table / ecdsa_ladder): the tables are built
// beside ecdsa_prep (FnEcdsaPrepTable), the ladder after both
template <class CV, bool WIDE = true>
struct FnEcdsaTable {
  static constexpr const char* NAME = "ecdsa_table";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = ELL_ECDSA_TABLE_MIN_WAVES;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* pub; typename W::VT* tbl;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::template ecdsa_table<WIDE>(i, n, pub, tbl);
  }
};
template <class CV, bool WIDE = true>
struct FnEcdsaLadder {
  static constexpr const char* NAME = "ecdsa_main";      // the timing name of pass 2 in either form
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = WIDE ? 3 : ELL_ENDO_MIN_WAVES;
  static constexpr int DS_PER_LANE = W::NWIN * W::NSV;
  size_t n; const u32* u12; const u8* valid; const u8* r; const u8* pub;
  const typename W::A* comb; const typename W::VT* tbl; u8* ok; u8* st;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) W::template ecdsa_ladder<WIDE>(i, n, u12, valid, r, pub, comb, tbl, ds, ok, st);
  }
};

// ecdsa_prep and ecdsa_table in ONE launch (horizontal fusion): workgroups below `tpad` threads
// run the prep (the 104-register instantiation: at 216 registers two waves of prep would leave a
// SIMD no room for the table workgroups), the others build window tables -- the two jobs share the
// SIMDs without a second stream, without events and without a cross-queue wait before the ladder
template <class CV>
struct FnEcdsaPrepTable {
  static constexpr const char* NAME = "ecdsa_prep_table";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = ELL_ECDSA_TABLE_MIN_WAVES;
  static constexpr int DS_PER_LANE = 0;
  FnEcdsaPrep<CV, ELL_ECDSA_TABLE_MIN_WAVES> prep; size_t tpad; FnEcdsaTable<CV, true> table;
  ELL_HD void operator()(size_t tid, const DigitStore& ds) const {
    if (tid < tpad) {
      if (fill_lane(tid, prep.T)) prep(tid, ds);
    } else {
      table(tid - tpad, ds);                     // (the launch ends with this range: k_run fills its last wave)
    }
  }
};

// The parted verify (Work::ecdsa_half / ecdsa_fixed / ecdsa_join): batches that leave most SIMDs
// without a wave.  One launch of three regions of `npad` threads -- whole workgroups each, so a
// wave belongs to ONE part -- runs the two half ladders and the comb of every item in different
// waves; the join adds up.  256 registers per lane: a lone chain has no neighbour to make room for.
template <class CV>
struct FnEcdsaParts {
  static constexpr const char* NAME = "ecdsa_parts";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = 2;
  static constexpr int DS_PER_LANE = W::template Endo<true>::NW;
  size_t n; size_t npad; const u32* u12; const typename W::A* comb; const typename W::VT* tbl; u32* jac;
  ELL_HD void operator()(size_t tid, const DigitStore& ds) const {
    const int part = tid >= 2 * npad ? 2 : (tid >= npad ? 1 : 0);
    size_t i = tid - (size_t)part * npad;
    if (!fill_lane(i, n)) return;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 2) W::ecdsa_fixed(i, n, u12, comb, out);
    else W::template ecdsa_half<true>(i, n, part, u12, tbl, ds, out);
  }
};
template <class CV>
struct FnEcdsaJoin {
  static constexpr const char* NAME = "ecdsa_join";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = 2;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* valid; const u8* r; const u8* pub; const typename W::VT* tbl; const u32* jac; u8* ok; u8* st;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::template ecdsa_join<true>(i, n, valid, r, pub, tbl, jac, ok, st);
  }
};

// The parts of the parted verify / Point#mul on the LANES-PER-ITEM layer (coop.h, coop_work.h):
// one unit = one part of one item on a wave of its own, its field elements spread over a 16-lane
// row -- 2.2 x fewer instructions on the item's critical path.  Launched through BK::launch_coop
// (k_run_coop: one unit per workgroup) for batches of at most Tuning::coop_grid items; the join
// kernels are the one-lane ones.  secp256k1 only.
struct FnEcdsaPartsC {
  static constexpr const char* NAME = "ecdsa_parts_c";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = CoopK256::E::NW;
  static constexpr int ROW_BYTES = CoopK256::ROW_BYTES;
  size_t n; const u32* u12; const typename W::A* comb; const typename W::VT* tbl; u32* jac;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const int part = (int)(unit / n);
    const size_t i = unit - (size_t)part * n;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 2) CoopK256::ecdsa_fixed(i, n, u12, comb, out);
    else CoopK256::ecdsa_half(i, n, part, u12, tbl, ds, out, row_mem);
  }
};
// ... and in front of them, ONE launch: unit i < n runs the scalar-field prep of item i (the
// one-lane code on a wave of its own: range checks, s^-1, u1, u2), unit n + i builds the window
// table of Q_i on the row layer (coop_work.h CoopK256::ecdsa_table) -- the one-lane
// FnEcdsaPrepTable took as long as its table build (86 us against the prep's 55).
struct FnEcdsaPrepTableC {
  static constexpr const char* NAME = "ecdsa_prep_table_c";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = CoopK256::ROW_BYTES;
  size_t n; const u8* hash; int hash_len; int shift; const u8* r; const u8* s;
  u32* pre; u32* u12; u8* valid; const u8* pub; typename W::VT* tbl;
  ELL_HD void operator()(size_t unit, const DigitStore&, void* row_mem) const {
    // (every lane of the prep's wave runs the one-lane code on the same item and stores the same values)
    if (unit < n) W::ecdsa_prep(unit, n, n, 1, hash, hash_len, shift, r, s, pre, u12, valid);
    else CoopK256::ecdsa_table(unit - n, pub, tbl, row_mem);
  }
};
struct FnMulPartsC {
  static constexpr const char* NAME = "mul_parts_c";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = CoopK256::E::NW;
  static constexpr int ROW_BYTES = CoopK256::ROW_BYTES;
  // kg != null: k*P + kg*G -- the units of a third range run the comb of kg
  size_t n; const u8* k; const u8* xy; u32* jac; const u8* kg; const typename W::A* comb;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const int part = (int)(unit / n);
    const size_t i = unit - (size_t)part * n;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 2) CoopK256::mul_fixed_part(i, n, kg, comb, out);
    else CoopK256::mul_half(i, n, part, k, xy, ds, out, row_mem);
  }
};

// The same parts with ONE ITEM PER ROW of the wave (coop.h FpK256R, coop_work.h CoopK256R): four
// items per unit, for batches between Tuning::coop_grid and Tuning::row_grid -- too many for a wave
// per part (that layer holds 1 365 verifies at four waves per SIMD), too few for the
// one-item-per-lane kernels, whose chain is 4.4x longer (one 128-bit ladder: 573 us on a lone
// one-lane wave, 130 us on the row layer; 4 096 verifies used 192 of the 1 024 SIMDs).
// Unit u of a part takes items 4 u .. 4 u + 3; rows past the end redo the last item.
ELL_HD size_t row_item(size_t group, int row, size_t n) {
  const size_t i = group * 4 + (size_t)row;
  return i < n ? i : n - 1;
}
struct FnEcdsaPartsR {
  static constexpr const char* NAME = "ecdsa_parts_r";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = CoopK256R::E::NW;
  static constexpr int ROW_BYTES = CoopK256R::ROW_BYTES;
  size_t n; const u32* u12; const typename W::A* comb; const typename W::VT* tbl; u32* jac;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const size_t groups = (n + 3) / 4;
    const int part = (int)(unit / groups);
    const size_t g = unit - (size_t)part * groups;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    ELL_FOR_ROWS(row) {
      const size_t i = row_item(g, row, n);
      if (part == 2) CoopK256R::ecdsa_fixed(i, n, u12, comb, out);
      else CoopK256R::ecdsa_half(i, n, part, u12, tbl, ds, out, row_mem);
    }
  }
};
// in front of them, ONE launch: the first units run the scalar-field prep one item per LANE (the
// one-lane code: 64 items per wave, one inversion each -- the chain counts), the others build the
// window tables of the keys four per wave
struct FnEcdsaPrepTableR {
  static constexpr const char* NAME = "ecdsa_prep_table_r";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = CoopK256R::ROW_BYTES;
  size_t n; size_t prep_units; const u8* hash; int hash_len; int shift; const u8* r; const u8* s;
  u32* pre; u32* u12; u8* valid; const u8* pub; typename W::VT* tbl;
  ELL_HD void operator()(size_t unit, const DigitStore&, void* row_mem) const {
    if (unit < prep_units) {
      ELL_FOR_WAVE_LANES(lane) {
        size_t t = unit * 64 + (size_t)lane;
        if (fill_lane(t, n)) W::ecdsa_prep(t, n, n, 1, hash, hash_len, shift, r, s, pre, u12, valid);
      }
    } else {
      ELL_FOR_ROWS(row) CoopK256R::ecdsa_table(row_item(unit - prep_units, row, n), pub, tbl, row_mem);
    }
  }
};
struct FnMulPartsR {
  static constexpr const char* NAME = "mul_parts_r";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = CoopK256R::E::NW;
  static constexpr int ROW_BYTES = CoopK256R::ROW_BYTES;
  size_t n; const u8* k; const u8* xy; u32* jac; const u8* kg; const typename W::A* comb;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const size_t groups = (n + 3) / 4;
    const int part = (int)(unit / groups);
    const size_t g = unit - (size_t)part * groups;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    ELL_FOR_ROWS(row) {
      const size_t i = row_item(g, row, n);
      if (part == 2) CoopK256R::mul_fixed_part(i, n, kg, comb, out);
      else CoopK256R::mul_half(i, n, part, k, xy, ds, out, row_mem);
    }
  }
};

// ... and for the NIST curves up to 256 bits (coop_mont.h, coop_work.h CoopNist): a verify is two
// units (u2*Q ladder, u1*G comb), a Point#mul one, k1*G + k2*P two
template <class CV>
struct FnEcdsaPartsN {
  static constexpr const char* NAME = "ecdsa_parts_c";
  typedef Work<CV> W;
  typedef CoopNist<CV> CW;
  static constexpr int DS_PER_LANE = CW::NW;
  static constexpr int ROW_BYTES = CW::ROW_BYTES;
  size_t n; const u32* u12; const u8* pub; const typename W::A* comb; u32* jac; const typename CW::A* gtbl;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const int part = (int)(unit / n);
    const size_t i = unit - (size_t)part * n;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 1) CW::ecdsa_fixed(i, n, u12, comb, out);
    else CW::ecdsa_var(i, n, u12, pub, ds, out, row_mem, gtbl);
  }
};
// ... in front of them, ONE launch: unit i < n runs the scalar-field prep of item i (the one-lane
// code on a wave), unit n + i builds the window table of Q_i -- co-Z chain, the inversion that maps
// it back to the curve, 60 us of the ladder's chain that do not depend on s^-1 -- and leaves it in
// global memory in the row's own format (CoopNist::ecdsa_table)
template <class CV>
struct FnEcdsaPrepTableN {
  static constexpr const char* NAME = "ecdsa_prep_table_c";
  typedef Work<CV> W;
  typedef CoopNist<CV> CW;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = CW::ROW_BYTES;
  size_t n; const u8* hash; int hash_len; int shift; const u8* r; const u8* s;
  u32* pre; u32* u12; u8* valid; const u8* pub; typename CW::A* gtbl;
  ELL_HD void operator()(size_t unit, const DigitStore&, void* row_mem) const {
    if (unit < n) W::ecdsa_prep(unit, n, n, 1, hash, hash_len, shift, r, s, pre, u12, valid);
    else CW::ecdsa_table(unit - n, pub, gtbl, row_mem);
  }
};
template <class CV>
struct FnMulPartsN {
  static constexpr const char* NAME = "mul_parts_c";
  typedef Work<CV> W;
  typedef CoopNist<CV> CW;
  static constexpr int DS_PER_LANE = CW::NW;
  static constexpr int ROW_BYTES = CW::ROW_BYTES;
  // kg != null: k*P + kg*G -- the units of a second range run the comb of kg
  size_t n; const u8* k; const u8* xy; u32* jac; const u8* kg; const typename W::A* comb;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const int part = (int)(unit / n);
    const size_t i = unit - (size_t)part * n;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 1) CW::mul_fixed_part(i, n, kg, comb, out);
    else CW::mul_var(i, n, k, xy, ds, out, row_mem);
  }
};
template <class CV>
struct FnEcdsaJoin2 {
  static constexpr const char* NAME = "ecdsa_join";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = 2;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* valid; const u8* r; const u8* pub; const u32* jac; u8* ok; u8* st;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::ecdsa_join2(i, n, valid, r, pub, jac, ok, st);
  }
};

// Point#mul in the parted form (Work::mul_half / mul_join)
template <class CV>
struct FnMulParts {
  static constexpr const char* NAME = "mul_parts";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = 2;
  static constexpr int DS_PER_LANE = W::template Endo<true>::NW;
  // kg != null: k*P + kg*G -- a third range of lanes runs the comb of kg
  size_t n; size_t npad; const u8* k; const u8* xy; typename W::VT* tbl; u32* jac;
  const u8* kg; const typename W::A* comb;
  ELL_HD void operator()(size_t tid, const DigitStore& ds) const {
    const int part = tid >= 2 * npad ? 2 : (tid >= npad ? 1 : 0);
    size_t i = tid - (size_t)part * npad;
    if (!fill_lane(i, n)) return;
    u32* out = jac + (size_t)part * 3 * W::NS * n;
    if (part == 2) W::mul_fixed_part(i, n, kg, comb, out);
    else W::template mul_half<true>(i, n, part, k, xy, tbl + (size_t)part * n * W::template stride<true>(), ds, out);
  }
};
template <class CV>
struct FnMulJoin {
  static constexpr const char* NAME = "mul_join";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = 2;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u32* jac; bool with_g; const u8* xy; u8* out_xy; u8* out_inf; typename W::A* raw;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::mul_join(i, n, jac, with_g, xy, out_xy, out_inf, raw);
  }
};

template <class CV, int MW = 0>
struct FnSignMul {
  static constexpr const char* NAME = "sign_mul";
  typedef Work<CV> W;
  static constexpr int MIN_WAVES = MW ? MW : (W::L <= 8 ? ELL_FIXED_MIN_WAVES : 1);
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* nonces; const typename W::A* comb; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::sign_mul(i, n, nonces, comb, jac);
  }
};
template <class CV>
struct FnSignFinish {
  static constexpr const char* NAME = "sign_finish";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u8* hash; int hash_len; int shift; const u8* priv;
  const u8* nonces; const u8* kg_xy; const u8* kg_inf; int canonical; u32* pre;
  u8* out_r; u8* out_s; u8* out_recid; u8* out_ok; const u32* kinv;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) W::sign_finish(t, T, n, K, hash, hash_len, shift, priv, nonces, kg_xy, kg_inf, canonical,
                              pre, out_r, out_s, out_recid, out_ok, kinv);
  }
};
// EC#sign's middle for a handful of items, ONE launch of the row layer (k_run_coop): unit i < n
// computes k_i*G on a wave of its own and leaves it affine (coop_work.h coop_sign_point), unit
// n + i inverts k_i mod n beside it (Work::sign_kinv) -- sign_mul, normalize and the inversion of
// sign_finish were three dependent one-lane launches.  CW: CoopK256 / CoopNist<CV>.
template <class CV, class CW>
struct FnSignPartsC {
  static constexpr const char* NAME = "sign_parts_c";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = 16;
  size_t n; const u8* nonces; const typename W::A* comb; u8* kg_xy; u8* kg_inf; u32* kinv;
  ELL_HD void operator()(size_t unit, const DigitStore&, void*) const {
    if (unit < n) coop_sign_point<CW>(unit, nonces, comb, kg_xy, kg_inf);
    else W::sign_kinv(unit - n, n, nonces, kinv, CW::writer());
  }
};
// Point#mul on G / KeyPair#getPublic for a handful of items on the row layer: the comb and the
// item's own inversion on a wave (coop_work.h coop_sign_point -- the point half of sign_parts_c);
// mul_fixed + normalize were two dependent one-lane launches.  The scalar is any BYTES-byte value
// (the comb's windows cover 8 BYTES bits and their carry, as in the one-lane mul_fixed).
template <class CV, class CW>
struct FnMulFixedC {
  static constexpr const char* NAME = "mul_fixed_c";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = 16;
  size_t n; const u8* k; const typename W::A* comb; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t unit, const DigitStore&, void*) const {
    if (unit < n) coop_sign_point<CW, false>(unit, k, comb, out_xy, out_inf);
  }
};
// ShortCurve#pointFromX of a handful of secp256k1 abscissas: the square root's 266 products on a
// wave per item (coop_work.h CoopK256::decompress); FnDecompress is the one-lane form.
struct FnDecompressC {
  static constexpr const char* NAME = "decompress_c";
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = 16;
  size_t n; const u8* xs; const u8* odd; u8* out_xy; u8* out_ok;
  ELL_HD void operator()(size_t unit, const DigitStore&, void*) const {
    if (unit < n) CoopK256::decompress(unit, xs, odd, out_xy, out_ok);
  }
};
// EC#recoverPubKey's front for a handful of items, ONE launch: unit i < n lifts R from (r, j) on
// a wave (the square root), unit n + i runs recover_prep for item i beside it (range checks, r^-1,
// both scalars) -- recover_prep and decompress were two dependent one-lane launches, 62 + 103 us.
struct FnRecoverPartsC {
  static constexpr const char* NAME = "recover_parts_c";
  typedef Work<CvSecp256k1> W;
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = 16;
  size_t n; const u8* hash; int hash_len; const u8* r; const u8* s; const u8* recid;
  u32* pre; u8* xs; u8* odd; u8* s1; u8* s2; u8* status; u8* rxy; u8* dec_ok;
  ELL_HD void operator()(size_t unit, const DigitStore&, void*) const {
    if (unit < n) CoopK256::recover_point(unit, r, recid, rxy, dec_ok);
    // (every lane of the wave runs the one-lane code on the same item and stores the same values)
    else W::recover_prep(unit - n, n, n, 1, hash, hash_len, r, s, recid, pre, xs, odd, s1, s2, status);
  }
};
template <class CV>
struct FnDetNonce {
  static constexpr const char* NAME = "det_nonce";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* hash; int hash_len; int shift; const u8* priv; u8* nonces;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::det_nonce(i, hash, hash_len, shift, priv, nonces);
  }
};
template <class CV>
struct FnRecoverPrep {
  static constexpr const char* NAME = "recover_prep";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t T, n; int K; const u8* hash; int hash_len; const u8* r; const u8* s; const u8* recid;
  u32* pre; u8* xs; u8* odd; u8* s1; u8* s2; u8* status;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) W::recover_prep(t, T, n, K, hash, hash_len, r, s, recid, pre, xs, odd, s1, s2, status);
  }
};
template <class CV>
struct FnRecoverFinish {
  static constexpr const char* NAME = "recover_finish";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* dec_ok; const u8* inf; u8* out_xy; u8* status;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::recover_finish(i, dec_ok, inf, out_xy, status);
  }
};
template <class CV>
struct FnDecompress {
  static constexpr const char* NAME = "decompress";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* x; const u8* odd; u8* out_xy; u8* out_ok;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if constexpr (CV::F::HAS_SQRT) { if (i < n) W::decompress(i, x, odd, out_xy, out_ok); }
  }
};
struct FnEdDecompress {
  static constexpr const char* NAME = "ed_decompress";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* y; const u8* odd; u8* out_xy; u8* out_ok;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::decompress(i, y, odd, out_xy, out_ok);
  }
};

template <class CV>
struct FnDecodePoint {
  static constexpr const char* NAME = "decode_point";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* enc; size_t len; u8* out_xy; u8* status;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::decode_point(i, enc, len, out_xy, status);
  }
};
template <class CV>
struct FnEncodePoint {
  static constexpr const char* NAME = "encode_point";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy; int compact; u8* out;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::encode_point(i, xy, compact, out);
  }
};
template <class CV>
struct FnValidatePoint {
  static constexpr const char* NAME = "validate_point";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy; const u8* inf; u8* status; u8* scal;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) {
      W::validate_point(i, xy, inf, status);
      if (scal) W::fill_order(i, scal);
    }
  }
};
template <class CV>
struct FnSigFromDer {
  static constexpr const char* NAME = "sig_from_der";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* der; size_t stride; const u32* der_len; u8* r; u8* s; u8* status;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::sig_from_der(i, der, stride, der_len, r, s, status);
  }
};
template <class CV>
struct FnSigToDer {
  static constexpr const char* NAME = "sig_to_der";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* r; const u8* s; u8* out; size_t stride; u32* out_len;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::sig_to_der(i, r, s, out, stride, out_len);
  }
};
template <class CV>
struct FnWireStatus {
  static constexpr const char* NAME = "wire_status";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* key_st; const u8* sig_st; const u8* ver_st; u8* ok; u8* err;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::wire_status(i, key_st, sig_st, ver_st, ok, err);
  }
};
template <class CV>
struct FnPointAdd {
  static constexpr const char* NAME = "point_add";
  typedef Work<CV> W;
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* inf1; const u8* xy2; const u8* inf2; u32* jac;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) W::point_add(i, n, xy1, inf1, xy2, inf2, jac);
  }
};
struct FnEdPointAdd {
  static constexpr const char* NAME = "ed_point_add";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* inf1; const u8* xy2; const u8* inf2; u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::point_add(i, n, xy1, inf1, xy2, inf2, ext);
  }
};
struct FnEdDecodePoint {
  static constexpr const char* NAME = "ed_decode_point";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* enc; u8* out_xy; u8* status;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::decode_points(i, enc, out_xy, status);
  }
};
struct FnEdEncodePoint {
  static constexpr const char* NAME = "ed_encode_point";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy; u8* out;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::encode_points(i, xy, out);
  }
};
struct FnEdValidatePoint {
  static constexpr const char* NAME = "ed_validate_point";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy; const u8* inf; u8* status; u8* scal;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) {
      EdWork::validate_point(i, xy, inf, status);
      if (scal) EdWork::fill_order(i, scal);
    }
  }
};
// third test of KeyPair#validate: status 0 stays 0 only when n * P came out as infinity
struct FnOrderStatus {
  static constexpr const char* NAME = "order_status";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* mul_inf; u8* status;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n && status[i] == 0 && !mul_inf[i]) status[i] = 3;
  }
};

struct FnEddsaVerify {
  static constexpr const char* NAME = "eddsa_verify";
  static constexpr int DS_PER_LANE = EdWork::NWIN;
  static constexpr int MIN_WAVES = 3;          // <= 168 VGPRs (unconstrained it takes 170: 2 waves/SIMD)
  size_t n; const u8* msgs; const u64* off; size_t msg_len; const u8* sigs; const u8* pubs;
  const EdWork::P* comb; EdWork::P* tbl; u8* ok; u8* err;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i >= n) return;
    const u8* m = off ? msgs + off[i] : msgs + i * msg_len;
    u64 len = off ? off[i + 1] - off[i] : (u64)msg_len;
    EdWork::eddsa_verify(i, m, len, sigs + i * 64, pubs + i * 32, comb, tbl, ds, ok, err);
  }
};

// EDDSA#verify for a handful of items: the two sides of the equation on a wave each (coop_ed.h),
// then the one-lane comparison
struct FnEddsaPartsC {
  static constexpr const char* NAME = "eddsa_parts_c";
  static constexpr int DS_PER_LANE = EdWork::NWIN;
  static constexpr int ROW_BYTES = CoopEd::ROW_BYTES2;      // (lane_table strides by sixteen entries)
  size_t n; const u8* msgs; const u64* off; size_t msg_len; const u8* sigs; const u8* pubs;
  const EdWork::P* comb; u32* ext; u8* flags;
  ELL_HD void operator()(size_t unit, const DigitStore& ds, void* row_mem) const {
    const int part = (int)(unit / n);
    const size_t i = unit - (size_t)part * n;
    const u8* m = off ? msgs + off[i] : msgs + i * msg_len;
    const u64 len = off ? off[i + 1] - off[i] : (u64)msg_len;
    CoopEd::verify_part(i, n, part, m, len, sigs + i * 64, pubs + i * 32, comb, ds, ext, flags, row_mem);
  }
};
// edwards Point#mul / mulAdd for a handful of items: one item per wave (coop_ed.h), then the
// one-lane normalisation
struct FnEdMulC {
  static constexpr const char* NAME = "ed_mul_c";
  static constexpr int DS_PER_LANE = 2 * EdWork::NWIN;
  static constexpr int ROW_BYTES = CoopEd::ROW_BYTES2;
  // k1 == null: k2*P2; xy1 == null: k1*G + k2*P2
  size_t n; const u8* k1; const u8* xy1; const u8* k2; const u8* xy2; const EdWork::P* comb; u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore& ds, void* row_mem) const {
    if (!k1) CoopEd::mul_var(i, n, k2, xy2, ds, ext, row_mem);
    else CoopEd::mul_add(i, n, k1, xy1, k2, xy2, comb, ds, ext, row_mem);
  }
};
// edwards Point#mul on G for a handful of items: the comb and the inversion on a wave, one launch
struct FnEdMulFixedC {
  static constexpr const char* NAME = "ed_mul_fixed_c";
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = 16;
  size_t n; const u8* k; const EdWork::P* comb; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t i, const DigitStore&, void*) const {
    if (i < n) CoopEd::mul_fixed(i, k, comb, out_xy, out_inf);
  }
};
struct FnX25519C {
  static constexpr const char* NAME = "x25519_c";
  static constexpr int DS_PER_LANE = 0;
  static constexpr int ROW_BYTES = CoopX25519::ROW_BYTES;
  // bad != null (KeyPair#derive): units n .. 2 n - 1 test the abscissas beside the ladders
  size_t n; const u8* k; const u8* x; u32* xz; u8* bad;
  ELL_HD void operator()(size_t i, const DigitStore&, void*) const {
    if (i < n) CoopX25519::ladder(i, n, k, x, xz);
    else CoopX25519::validate(i - n, x, bad);
  }
};
struct FnX25519Validate {
  static constexpr const char* NAME = "x25519_validate";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* x; u8* bad;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) MontWork::validate(i, x, bad);
  }
};
struct FnEddsaJoin {
  static constexpr const char* NAME = "eddsa_join";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u32* ext; const u8* flags; u8* ok; u8* err;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::eddsa_join(i, n, ext, flags, ok, err);
  }
};

// Edwards / Montgomery functors
struct FnEddsaSignPre {
  static constexpr const char* NAME = "eddsa_sign_pre";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* secrets; const u8* msgs; const u64* off; size_t msg_len; u8* scal;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i >= n) return;
    const u8* m = off ? msgs + off[i] : msgs + i * msg_len;
    u64 len = off ? off[i + 1] - off[i] : (u64)msg_len;
    EdWork::sign_pre(i, n, secrets + i * 32, m, len, scal);
  }
};
struct FnEddsaSignPost {
  static constexpr const char* NAME = "eddsa_sign_post";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* msgs; const u64* off; size_t msg_len; const u8* scal; const u8* xy;
  u8* sig; u8* pub;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i >= n) return;
    const u8* m = off ? msgs + off[i] : msgs + i * msg_len;
    u64 len = off ? off[i + 1] - off[i] : (u64)msg_len;
    EdWork::sign_post(i, n, m, len, scal, xy, sig, pub);
  }
};
struct FnEdMulVar {
  static constexpr const char* NAME = "ed_mul_var";
  static constexpr int DS_PER_LANE = EdWork::NWIN;
  static constexpr int MIN_WAVES = 4;          // <= 128 VGPRs
  size_t n; const u8* k; const u8* xy; EdWork::P* tbl; u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) EdWork::mul_var(i, n, k, xy, tbl, ds, ext);
  }
};
struct FnEdMulFixed {
  static constexpr const char* NAME = "ed_mul_fixed";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* k; const EdWork::P* comb; u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdWork::mul_fixed(i, n, k, comb, ext);
  }
};
struct FnEdMulAddG {
  static constexpr const char* NAME = "ed_mul_add_g";
  static constexpr int DS_PER_LANE = EdWork::NWIN;
  size_t n; const u8* k1; const u8* k2; const u8* xy2; const EdWork::P* comb; EdWork::P* tbl;
  u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) EdWork::mul_add_g(i, n, k1, k2, xy2, comb, tbl, ds, ext);
  }
};
struct FnEdMulAdd2 {
  static constexpr const char* NAME = "ed_mul_add2";
  static constexpr int DS_PER_LANE = EdWork::NWIN * 2;
  size_t n; const u8* k1; const u8* xy1; const u8* k2; const u8* xy2; EdWork::P* tbl; u32* ext;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) EdWork::mul_add2(i, n, k1, xy1, k2, xy2, tbl, ds, ext);
  }
};
struct FnEdNormalize {
  static constexpr const char* NAME = "ed_normalize";
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u32* ext; u32* pre; u8* out_xy; u8* out_inf; EdWork::P* raw;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) EdWork::normalize(t, T, n, K, ext, pre, out_xy, out_inf, raw);
  }
};
struct FnX25519 {
  static constexpr const char* NAME = "x25519_ladder";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* k; const u8* x; u32* xz;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) MontWork::ladder(i, n, k, x, xz);
  }
};
struct FnX25519Normalize {
  static constexpr const char* NAME = "x25519_normalize";
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u32* xz; u32* pre; u8* out_x; u8* out_inf;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) MontWork::normalize(t, T, n, K, xz, pre, out_x, out_inf);
  }
};

// user-defined Edwards curves (edcustom.h)
struct FnEdcMulVar {
  static constexpr const char* NAME = "edc_mul_var";
  static constexpr int DS_PER_LANE = EdcWork::NWIN;
  static constexpr int MIN_WAVES = 3;
  size_t n; const u8* k; const u8* xy; EdcWork::P* tbl; u32* out;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) EdcWork::mul_var(i, n, k, xy, tbl, ds, out);
  }
};
struct FnEdcMulAdd2 {
  static constexpr const char* NAME = "edc_mul_add2";
  static constexpr int DS_PER_LANE = EdcWork::NWIN * 2;
  static constexpr int MIN_WAVES = 3;
  size_t n; const u8* k1; const u8* xy1; const u8* k2; const u8* xy2; EdcWork::P* tbl; u32* out;
  ELL_HD void operator()(size_t i, const DigitStore& ds) const {
    if (i < n) EdcWork::mul_add2(i, n, k1, xy1, k2, xy2, tbl, ds, out);
  }
};
struct FnEdcPointAdd {
  static constexpr const char* NAME = "edc_point_add";
  static constexpr int DS_PER_LANE = 0;
  size_t n; const u8* xy1; const u8* inf1; const u8* xy2; const u8* inf2; u32* out;
  ELL_HD void operator()(size_t i, const DigitStore&) const {
    if (i < n) EdcWork::point_add(i, n, xy1, inf1, xy2, inf2, out);
  }
};
struct FnEdcNormalize {
  static constexpr const char* NAME = "edc_normalize";
  static constexpr int DS_PER_LANE = 0;
  size_t T; size_t n; int K; const u32* proj; u32* pre; u8* out_xy; u8* out_inf;
  ELL_HD void operator()(size_t t, const DigitStore&) const {
    if (t < T) EdcWork::normalize(t, T, n, K, proj, pre, out_xy, out_inf);
  }
};

// ---- curve metadata ----------------------------------------------------------
struct CurveInfo {
  const char* name;
  int field_bytes;
  int order_bytes;
  int order_bits;
};
inline const CurveInfo* curve_info(int id) {
  static const CurveInfo tab[CURVE_COUNT] = {
      {"secp256k1", 32, 32, 256}, {"p192", 24, 24, 192}, {"p224", 28, 28, 224},
      {"p256", 32, 32, 256},      {"p384", 48, 48, 384}, {"p521", 66, 66, 521},
      {"ed25519", 32, 32, 253},   {"curve25519", 32, 32, 253}};
  // user-defined short curves (Engine::define_short): every width is 32 bytes whatever the prime's size
  static const CurveInfo custom = {"custom", 32, 32, 256};
  if (id >= CURVE_CUSTOM0 && id < CURVE_CUSTOM0 + CURVE_CUSTOM_MAX) return &custom;
  if (id < 0 || id >= CURVE_COUNT) return nullptr;
  return &tab[id];
}

inline bool is_custom(int curve) { return curve >= CURVE_CUSTOM0 && curve < CURVE_CUSTOM0 + CURVE_CUSTOM_MAX; }
// The custom-curve kernels read their modulus from ONE parameter block per device (fp_rt.h):
// calls on user-defined curves are serialised per device, from the upload of the block to the
// end of the call's device work.
inline std::mutex& custom_mutex(int device) {
  static std::mutex m[64];
  return m[(unsigned)device % 64u];
}

// ---- the engine ----------------------------------------------------------------
template <class BK>
class Engine {
 public:
  BK bk;
  std::string err;
  // items per field inversion (Montgomery's trick).  With the division-step inversion the
  // per-item part dominates a thread's work, so the scalar (mod n) kernels, which have nothing
  // else to hide their latency behind, prefer half the batch and twice the wavefronts
  // (ecdsa_prep 0.41 -> 0.34 ms, sign_finish 0.45 -> 0.30 ms per 2^20); normalize is best at 16.
  static constexpr int INV_BATCH = ELL_INV_BATCH;
  static constexpr int INV_BATCH_N = ELL_INV_BATCH / 2;
  // Items per inversion in the scalar-field kernels (ecdsa_prep ...): kmax for batches that fill
  // the device anyway; for smaller ones fewer items per thread, so that the n / K threads still
  // give every SIMD a wave (one inversion costs about what two items' other work does; a batch
  // of 131 072 verifies -- one GPU's share of 2^20 over eight -- ran its prep on 256 waves, one
  // per CU, at the lone-wave issue rate: 0.137 ms of a 1.52 ms pass).  ELLGPU_PREP_K overrides.
  // Batch-size dependent tuning, fixed when the context is created: the device's geometry comes
  // from the backend (compute units of THIS device, not a constant), the developer / test
  // overrides from the environment, read once here -- never at launch time:
  //   ELLGPU_SMALL_GRID    largest batch that takes the small-grid (WIDE) kernels (0 = never);
  //                        default: three waves on every SIMD of the device
  //   ELLGPU_SPLIT_VERIFY  0 keeps small-grid verifies on prep -> ecdsa_main (default: prep ||
  //                        table -> ladder)
  //   ELLGPU_PARTED_GRID   largest batch that takes the parted verify (FnEcdsaParts; 0 = never);
  //                        default: half a wave round -- the two half ladders of every item find a
  //                        SIMD of their own (the comb's waves are short)
  //   ELLGPU_PREP_K        items per inversion in the scalar-field kernels (default: by batch size)
  struct Tuning {
    size_t wave_round;        // lanes of one wave on every SIMD of the device
    size_t small_grid;
    bool split_verify;
    size_t parted_grid;       // largest batch that takes the parted verify (three lanes per item)
    size_t coop_grid;         // largest batch whose parts run on the lanes-per-item layer (a wave per part)
    size_t wide_grid;         // ... of p384 / p521 (one item per wave on the wide layer)
    size_t row_from, row_grid;  // batches of row_from < n <= row_grid items run their parts one item per ROW (four items per wave)
    size_t comb_max_bytes;    // ELLGPU_COMB_MAX_BYTES: fixed-base tables above this are treated as unallocatable (0 = no limit)
    int prep_k;
    int norm_k;
  };
  Tuning tune_;
  void init_tuning() {
    tune_.wave_round = (size_t)bk.compute_units() * 4 * 64;
    const char* e = getenv("ELLGPU_SMALL_GRID");
    tune_.small_grid = e ? (size_t)strtoull(e, nullptr, 10) : 3 * tune_.wave_round;
    e = getenv("ELLGPU_SPLIT_VERIFY");
    tune_.split_verify = !(e && e[0] == '0');
    e = getenv("ELLGPU_PARTED_GRID");
    tune_.parted_grid = e ? (size_t)strtoull(e, nullptr, 10) : tune_.wave_round / 2;
    // ELLGPU_COOP_GRID  largest batch whose parts take a WAVE each (coop.h; 0 = never); default:
    //                   four waves per SIMD's worth of verify parts (1 365 items on 256 CUs) --
    //                   measured (profiles/r05_small_batch_forms.jsonl): a pass of 1 365 verifies
    //                   0.64 ms on waves against 0.72 on lanes, of 2 048 0.75 against 0.73; above
    //                   that the waves share SIMDs and the one-lane parts (64 items per wave) win
    e = getenv("ELLGPU_COOP_GRID");
    tune_.coop_grid = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)bk.compute_units() * 4 * 4 / 3;
    // One item per row (four items per wave, a wave 1.3x as long as the one-item wave): measured
    // (profiles/r06_latency_rows_ab.txt) it passes the wave-per-part form at ~2.5 items per CU -- 768
    // verifies 382 -> 310 us -- and holds against the one-lane parted form's single 573-us chain up to
    // ~18 per CU: 4 096 verifies 768 -> 608 us, 4 778 a tie.  ELLGPU_ROW_FROM / ELLGPU_ROW_GRID.
    e = getenv("ELLGPU_WIDE_GRID");
    tune_.wide_grid = e ? (size_t)strtoull(e, nullptr, 10) : tune_.coop_grid;
    e = getenv("ELLGPU_ROW_FROM");
    tune_.row_from = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)bk.compute_units() * 5 / 2;
    e = getenv("ELLGPU_ROW_GRID");
    tune_.row_grid = e ? (size_t)strtoull(e, nullptr, 10) : (size_t)bk.compute_units() * 18;
    e = getenv("ELLGPU_COMB_MAX_BYTES");
    tune_.comb_max_bytes = e ? (size_t)strtoull(e, nullptr, 10) : 0;
    e = getenv("ELLGPU_PREP_K");
    tune_.prep_k = e ? atoi(e) : 0;
    e = getenv("ELLGPU_NORM_K");
    tune_.norm_k = e ? atoi(e) : 0;
  }
  size_t small_grid() const { return tune_.small_grid; }
  bool split_small_verify() const { return tune_.split_verify; }
  size_t parted_grid() const { return tune_.parted_grid; }
  size_t coop_grid() const { return CoopK256::AVAILABLE ? tune_.coop_grid : 0; }
  size_t row_grid() const { return CoopK256::AVAILABLE ? tune_.row_grid : 0; }
  // ... of curve CV: p384 / p521 (the wide layer, coop_wide.h) have a threshold of their own
  // (ELLGPU_WIDE_GRID; default: coop_grid)
  template <class CV>
  size_t coop_grid_of() const {
    if constexpr (!CV::ENDO && CoopConsts<CV>::WIDE) return tune_.wide_grid;
    else return coop_grid();
  }
  // do the parts of a secp256k1 verify / Point#mul of n items run one item per row?  (else one item per
  // wave up to coop_grid, one item per lane above)
  bool rows_for(size_t n) const { return CoopK256::AVAILABLE && n > tune_.row_from && n <= tune_.row_grid; }
  // window width of the curve's fixed-base table in use (0 = not built; ellgpu_ctx_comb_bits)
  int comb_bits(int curve) const {
    if (curve < 0 || curve >= CURVE_COUNT || !comb_[curve]) return 0;
    return comb_bits_[curve] ? comb_bits_[curve] : 8;
  }
  // ... and for the ecdsa_prep that runs BESIDE ecdsa_table (small-grid verify): the two kernels
  // share the SIMDs, so what counts is the work, not the latency of a lone chain -- the largest K
  // that still leaves half a wave round of threads (131 072 items: K = 4, 1.332 -> 1.318 ms per
  // pass; 65 536: K = 2; 16 384: K = 1; profiles/r04_split_verify_ab.txt)
  int inv_batch_beside(size_t n, int kmax) const {
    if (tune_.prep_k >= 1 && tune_.prep_k <= 64) return tune_.prep_k;
    int k = kmax;
    while (k > 1 && n / (size_t)k < tune_.wave_round / 2) k >>= 1;
    return k;
  }
  // items per inversion in normalize (ELLGPU_NORM_K overrides)
  // -- on a batch that leaves SIMDs idle the chain of K items per thread counts, not the inversions
  int norm_batch_for(size_t n) const {
    if (tune_.norm_k >= 1 && tune_.norm_k <= 64) return tune_.norm_k;
    int k = INV_BATCH;
    while (k > 1 && n / (size_t)k < tune_.wave_round / 2) k >>= 1;
    return k;
  }
  int inv_batch_for(size_t n, int kmax) const {
    if (tune_.prep_k >= 1 && tune_.prep_k <= 64) return tune_.prep_k;
    int k = kmax;
    while (k > 1 && n / (size_t)k < 2 * tune_.wave_round) k >>= 1;
    return k;
  }
  static constexpr size_t CHUNK = 1u << 21;   // max items per launch (bounds the scratch arena)

  explicit Engine(const BK& b) : bk(b) {
    for (int i = 0; i < CURVE_COUNT; i++) { comb_[i] = nullptr; comb_base_[i] = nullptr; comb_bits_[i] = 0; }
    init_tuning();
  }
  ~Engine() {
    for (int i = 0; i < CURVE_COUNT; i++)
      if (comb_[i]) bk.free_(comb_base_[i] ? comb_base_[i] : comb_[i]);
    for (auto& a : scratch_)
      for (auto& s : a)
        if (s.p) bk.free_(s.p);
    for (auto& s : staging_)
      if (s.p) bk.free_(s.p);
  }

  // ---- scratch arena: a few grow-only device buffers ----------------------
  struct Buf { void* p = nullptr; size_t cap = 0; };
  enum { S_TBL = 0, S_JAC, S_PRE, S_U12, S_VALID, S_WIRE, S_COUNT };
  enum { G_IN0 = 0, G_IN1, G_IN2, G_IN3, G_IN4, G_OUT0, G_OUT1, G_OUT2, G_COUNT };

  void* grow(Buf& b, size_t bytes) {
    if (bytes <= b.cap) return b.p;
    if (b.p) { bk.sync_all(); bk.free_(b.p); b.p = nullptr; b.cap = 0; }
    size_t want = bytes + bytes / 8;
    b.p = bk.alloc(want);
    if (!b.p) { b.p = bk.alloc(bytes); want = bytes; }
    if (!b.p) return nullptr;
    b.cap = want;
    return b.p;
  }
  void* scratch(int which, size_t bytes) { return grow(scratch_[lane_][which], bytes); }
  void* staging(int which, size_t bytes) { return grow(staging_[which], bytes); }

  int fail(int code, const char* msg) { err = msg; return code; }

  // ---- fixed-base comb tables, built on the device with our own kernels ----
  template <class CV>
  int ensure_comb();
  template <class CV>
  int normalize_chunk(size_t n, const u32* jac, u8* out_xy, u8* out_inf, typename Work<CV>::A* raw);
  template <class CV>
  int mul_var_chunk(size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf,
                    typename Work<CV>::A* raw);
  template <class CV>
  int mul_fixed_chunk(size_t n, const u8* k, u8* out_xy, u8* out_inf);
  template <class CV>
  int mul_add2_chunk(size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                     u8* out_xy, u8* out_inf);
  template <class CV>
  int mul_add_g_chunk(size_t n, const u8* k1, const u8* k2, const u8* xy2, u8* out_xy,
                      u8* out_inf);
  template <class CV>
  int ecdsa_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* r, const u8* s,
                  const u8* pub, u8* ok, u8* st);
  // launch of a scalar-field kernel (ecdsa_prep, sign_finish, recover_prep: batched inversion
  // and Montgomery arithmetic mod n, long dependent chains).  A member of its own so that these
  // kernels are instantiated in their own translation units (inst.hip group 6), which are
  // compiled with LLVM's max-ILP scheduling strategy: measured ecdsa_prep 0.334 -> 0.234 ms per
  // 2^20, while the ladder kernels are 0.4-1.3 % slower under it and keep the default.
  template <class Fn>
  int launch_fn(const Fn& f, size_t nthreads);
  template <int U = 0>
  int ensure_ed_comb();
  template <int U = 0>
  int ed_normalize_chunk(size_t n, const u32* ext, u8* out_xy, u8* out_inf, EdWork::P* raw);
  template <int U = 0>
  int ed_mul_var_chunk(size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf, EdWork::P* raw);
  template <int U = 0>
  int ed_mul_fixed_chunk(size_t n, const u8* k, u8* out_xy, u8* out_inf);
  template <int U = 0>
  int ed_mul_add2_chunk(size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                        u8* out_xy, u8* out_inf);
  template <int U = 0>
  int x25519_chunk(size_t n, const u8* k, const u8* x, u8* out_x, u8* out_inf, u8* out_bad);
  // user-defined Edwards curves: op 0 = k*P, 1 = k1*P1 + k2*P2, 2 = P1 + P2 (a, b = the inf flags)
  template <int U = 0>
  int edc_chunk(int op, size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                const u8* a, const u8* b, u8* out_xy, u8* out_inf);
  template <class CV>
  int decompress_chunk(size_t n, const u8* x, const u8* odd, u8* out_xy, u8* out_ok);
  template <class CV>
  int codec_chunk(int op, size_t n, const u8* in, size_t len, int flag, const u8* inf, u8* out, u8* status);
  template <class CV>
  int point_add_chunk(size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2, u8* out_xy, u8* out_inf);
  template <int U = 0>
  int ed_point_add_chunk(size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2, u8* out_xy, u8* out_inf);
  template <class CV>
  int der_chunk(int op, size_t n, const u8* a, const u8* b, size_t stride, u32* lens, u8* o1, u8* o2, u8* o3);
  template <class CV>
  int sign_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* priv, const u8* nonces,
                 int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok);
  template <class CV>
  int sign_det_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* priv, int canonical,
                     u8* out_r, u8* out_s, u8* out_recid, u8* out_ok);
  template <class CV>
  int recover_chunk(size_t n, const u8* hash, int hash_len, const u8* r, const u8* s, const u8* recid,
                    u8* out_xy, u8* out_status);
  template <int U = 0>
  int ed_decompress_chunk(size_t n, const u8* y, const u8* odd, u8* out_xy, u8* out_ok);
  template <int U = 0>
  int ed_codec_chunk(int op, size_t n, const u8* in, int flag, const u8* inf, u8* out, u8* status);
  template <int U = 0>
  int eddsa_chunk(size_t n, size_t o, const u8* msgs, const u64* off, size_t msg_len, const u8* sigs,
                  const u8* pubs, u8* ok, u8* err);
  template <int U = 0>
  int eddsa_sign_chunk(size_t n, size_t o, const u8* secrets, const u8* msgs, const u64* off,
                       size_t msg_len, u8* sig, u8* pub);

  // ---- dispatch over curves (device pointers) --------------------------------
#if defined(ELL_ONLY_CURVE)
// developer build (elliptic_amd/build.py --dev <curve>): one short curve's kernels only, for
// fast kernel iteration; every other short curve reports E_UNSUPPORTED
#define ELL_SHORT_DISPATCH(curve, CALL)                         \
  switch (curve) {                                              \
    case ELL_ONLY_CURVE: { typedef ELL_ONLY_TYPE CV; CALL; } break; \
    default: return fail(E_UNSUPPORTED, is_custom(curve) ? "not available on user-defined curves (scalar multiplication and point addition only)" : "developer build: single curve only"); \
  }
#else
#define ELL_SHORT_DISPATCH(curve, CALL)                         \
  switch (curve) {                                              \
    case CURVE_SECP256K1: { typedef CvSecp256k1 CV; CALL; } break; \
    case CURVE_P192: { typedef CvP192 CV; CALL; } break;           \
    case CURVE_P224: { typedef CvP224 CV; CALL; } break;           \
    case CURVE_P256: { typedef CvP256 CV; CALL; } break;           \
    case CURVE_P384: { typedef CvP384 CV; CALL; } break;           \
    case CURVE_P521: { typedef CvP521 CV; CALL; } break;           \
    default: return is_custom(curve) ? fail(E_UNSUPPORTED, "not available on user-defined curves (scalar multiplication and point addition only)") \
                                     : fail(E_ARG, "unknown curve id");            \
  }
#endif

  // ---- user-defined short Weierstrass curves (run-time prime, arbitrary a) ----------------
  // `new elliptic.curve.short({p, a, b})` (short.js:11-24) with parameters that are no preset:
  // p an odd prime < 2^256 (primality is the caller's business, as it is the reference's), a and
  // b any residues.  Registers the curve under an id >= CURVE_CUSTOM0 of this context; Point#mul,
  // mulAdd / jmulAdd and Point#add then run on the device through the generic-a doubling
  // (ShortOps::dbl, A_KIND 1) and a Montgomery field whose modulus is a kernel-time constant.
  // modulus-dependent constants of a parameter block
  int rt_field_init(RtField& f, const u8* p_be) {
    memset(&f, 0, sizeof(f));
    load_be<8>(f.p, p_be, 32);
    if (!(f.p[0] & 1u)) return fail(E_ARG, "user-defined curve: the modulus must be odd");
    bool small = true;
    for (int i = 1; i < 8; i++) small = small && f.p[i] == 0;
    if (small && f.p[0] < 5) return fail(E_ARG, "user-defined curve: the modulus must be a prime > 3");
    u32 inv = 1;                                          // p^-1 mod 2^32 (Newton)
    for (int i = 0; i < 5; i++) inv *= 2u - f.p[0] * inv;
    f.n0 = 0u - inv;
    u32 one[8];
    bn_zero<8>(one);
    one[0] = 1;
    rt_to_mont(f, f.one, one);
    bn_copy<8>(f.r2, f.one);
    rt_times_r(f, f.r2);
    u32 two[8];
    bn_zero<8>(two);
    two[0] = 2;
    bn_sub<8>(f.pm2, f.p, two);
    return E_OK;
  }
  static void rt_times_r(const RtField& f, u32 (&r)[8]) {          // r * 2^256 mod p, r < p
    for (int i = 0; i < 256; i++) {
      u32 t[8];
      mod_add<8>(t, r, r, f.p);
      bn_copy<8>(r, t);
    }
  }
  // out = x * 2^256 mod p for any x < 2^256 (reduced bit by bit first)
  static void rt_to_mont(const RtField& f, u32 (&out)[8], const u32 (&x)[8]) {
    u32 r[8];
    bn_zero<8>(r);
    for (int i = 255; i >= 0; i--) {
      u32 t[8];
      mod_add<8>(t, r, r, f.p);
      bn_copy<8>(r, t);
      if ((x[i >> 5] >> (i & 31)) & 1u) {
        u32 o[8];
        bn_zero<8>(o);
        o[0] = 1;
        mod_add<8>(t, r, o, f.p);
        bn_copy<8>(r, t);
      }
    }
    rt_times_r(f, r);
    bn_copy<8>(out, r);
  }
  int register_custom(const RtField& f, int* out_curve) {
    for (size_t i = 0; i < custom_.size(); i++)
      if (memcmp(&custom_[i], &f, sizeof(f)) == 0) { *out_curve = CURVE_CUSTOM0 + (int)i; return E_OK; }
    if ((int)custom_.size() >= CURVE_CUSTOM_MAX)
      return fail(E_UNSUPPORTED, "at most 16 user-defined curves per context");
    custom_.push_back(f);
    *out_curve = CURVE_CUSTOM0 + (int)custom_.size() - 1;
    return E_OK;
  }
  // parameter block of a user-defined curve (edwards = 0: short Weierstrass, b; 1: Edwards, d)
  int build_custom(int edwards, const u8* p_be, const u8* a_be, const u8* bd_be, RtField& f) {
    if (!p_be || !a_be || !bd_be) return fail(E_ARG, "null pointer");
    int rc = rt_field_init(f, p_be);
    if (rc) return rc;
    u32 a[8], b[8];
    load_be<8>(a, a_be, 32);
    load_be<8>(b, bd_be, 32);
    rt_to_mont(f, f.a_m, a);
    if (!edwards) {
      rt_to_mont(f, f.b_m, b);
      u32 three[8], m3[8];
      bn_zero<8>(three);
      three[0] = 3;
      bn_sub<8>(m3, f.p, three);
      rt_to_mont(f, m3, m3);
      f.a_kind = bn_is_zero<8>(f.a_m) ? 0u : (bn_eq<8>(f.a_m, m3) ? 3u : 1u);
      f.kind = 0;
    } else {
      rt_to_mont(f, f.d_m, b);
      if (bn_is_zero<8>(f.a_m) || bn_is_zero<8>(f.d_m) || bn_eq<8>(f.a_m, f.d_m))
        return fail(E_ARG, "user-defined Edwards curve: a and d must be non-zero and distinct");
      f.kind = 1;
    }
    return E_OK;
  }
  int define_short(const u8* p_be, const u8* a_be, const u8* b_be, int* out_curve) {
    if (!out_curve) return fail(E_ARG, "null pointer");
    RtField f;
    int rc = build_custom(0, p_be, a_be, b_be, f);
    return rc ? rc : register_custom(f, out_curve);
  }
  // `new elliptic.curve.edwards({p, a, c: 1, d})` (edwards.js:11-31) with parameters that are not
  // ed25519's: a x^2 + y^2 = 1 + d x^2 y^2 over an odd prime p < 2^256; Point#mul, mulAdd and
  // Point#add run on the device in projective coordinates (edcustom.h).
  int define_edwards(const u8* p_be, const u8* a_be, const u8* d_be, int* out_curve) {
    if (!out_curve) return fail(E_ARG, "null pointer");
    RtField f;
    int rc = build_custom(1, p_be, a_be, d_be, f);
    return rc ? rc : register_custom(f, out_curve);
  }
  // the id a definition WOULD get (the existing one for parameters already registered, else the
  // next free one), -1 when the table is full; registers nothing.  Invalid parameters report the
  // next free id: the definition itself then fails with its own message, on the first member.
  int custom_slot_for(int edwards, const u8* p_be, const u8* a_be, const u8* bd_be) {
    RtField f;
    std::string keep = err;
    int rc = build_custom(edwards, p_be, a_be, bd_be, f);
    err = keep;
    if (rc == E_OK)
      for (size_t i = 0; i < custom_.size(); i++)
        if (memcmp(&custom_[i], &f, sizeof(f)) == 0) return CURVE_CUSTOM0 + (int)i;
    if ((int)custom_.size() >= CURVE_CUSTOM_MAX) return rc == E_OK ? -1 : CURVE_CUSTOM0 + (int)custom_.size();
    return CURVE_CUSTOM0 + (int)custom_.size();
  }
  static size_t pipe_step_default() {
    const char* e = getenv("ELLGPU_PIPE_STEP");          // developer override
    int v = e ? atoi(e) : ELL_PIPE_STEP_DEFAULT;
    return (size_t)(v >= 1 && v <= 64 ? v : ELL_PIPE_STEP_DEFAULT);
  }
  bool custom_is_edwards(int curve) const {
    size_t slot = (size_t)(curve - CURVE_CUSTOM0);
    return is_custom(curve) && slot < custom_.size() && custom_[slot].kind == 1;
  }
  // Brackets one call on a user-defined curve: takes the device's custom-curve lock, uploads the
  // curve's parameter block (synchronously: every earlier user of the block has finished, see the
  // destructor) and, at the end, waits for the call's device work before the lock is released.
  // Nested uses (the host-buffer wrappers call the *_dev forms chunk by chunk) are no-ops.
  struct CustomScope {
    Engine* e;
    bool owner = false;
    int rc = E_OK;
    CustomScope(Engine* eng, int curve) : e(eng) {
      if (!is_custom(curve) || e->custom_active_) return;
      size_t slot = (size_t)(curve - CURVE_CUSTOM0);
      if (slot >= e->custom_.size()) { rc = e->fail(E_ARG, "unknown curve id"); return; }
      custom_mutex(e->bk.device_index()).lock();
      owner = true;
      e->custom_active_ = true;
      e->bk.rt_upload(e->custom_[slot]);
    }
    ~CustomScope() {
      if (!owner) return;
      e->bk.sync();
      e->custom_active_ = false;
      custom_mutex(e->bk.device_index()).unlock();
    }
  };

  int prepare_curve(int curve) {
#if defined(ELL_ONLY_CURVE)
    if (curve < CURVE_ED25519 && curve != ELL_ONLY_CURVE) return fail(E_UNSUPPORTED, "developer build: single curve only");
#endif
    if (curve == CURVE_ED25519) return ensure_ed_comb();
    if (curve == CURVE_CURVE25519) return E_OK;
    int rc = E_OK;
    ELL_SHORT_DISPATCH(curve, rc = ensure_comb<CV>());
    return rc;
  }

  int mul_fixed_dev(int curve, size_t n, const u8* k, u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519)
      return fail(E_UNSUPPORTED, "curve25519 has no affine fixed-base form; use x25519_ladder");
    if (n && (!k || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    const size_t B = ci->field_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      if (curve == CURVE_ED25519) rc = ed_mul_fixed_chunk(m, k + o * B, out_xy + o * 2 * B, out_inf + o);
      else ELL_SHORT_DISPATCH(curve, rc = mul_fixed_chunk<CV>(m, k + o * B, out_xy + o * 2 * B, out_inf + o));
      if (rc) return rc;
    }
    return E_OK;
  }

  static bool overlap(const u8* a, const u8* b, size_t bytes) {
    return a && b && bytes && (uintptr_t)a < (uintptr_t)b + bytes && (uintptr_t)b < (uintptr_t)a + bytes;
  }
  int mul_var_dev(int curve, size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519)
      return fail(E_UNSUPPORTED, "curve25519 is x-only; use x25519_ladder");
    if (n && (!k || !xy || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    // the operands are read again AFTER the results are written (the curve test of
    // Work::domain_mark / mul_join): a result written over its own operand would be tested in its place
    if (overlap(xy, out_xy, n * 2 * B)) return fail(E_ARG, "in_xy and out_xy must not overlap");
    int rc = E_OK;
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      if (custom_is_edwards(curve))
        rc = edc_chunk(0, m, k + o * B, xy + o * 2 * B, nullptr, nullptr, nullptr, nullptr, out_xy + o * 2 * B, out_inf + o);
      else if (is_custom(curve))
        rc = mul_var_chunk<CvCustom>(m, k + o * B, xy + o * 2 * B, out_xy + o * 2 * B, out_inf + o, nullptr);
      else if (curve == CURVE_ED25519)
        rc = ed_mul_var_chunk(m, k + o * B, xy + o * 2 * B, out_xy + o * 2 * B, out_inf + o, nullptr);
      else
        ELL_SHORT_DISPATCH(curve, rc = mul_var_chunk<CV>(m, k + o * B, xy + o * 2 * B,
                                                         out_xy + o * 2 * B, out_inf + o, nullptr));
      if (rc) return rc;
    }
    return E_OK;
  }

  int mul_add2_dev(int curve, size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                   u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519)
      return fail(E_UNSUPPORTED, "Not supported on Montgomery curve");     // mont.js:155-157
    if (n && (!k1 || !k2 || !xy2 || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    int rc = E_OK;
    if (is_custom(curve) && !xy1)
      return fail(E_UNSUPPORTED, "user-defined curves have no fixed-base table: pass the generator as p1");
    if (!xy1) { rc = prepare_curve(curve); if (rc) return rc; }
    const size_t B = ci->field_bytes;
    if (overlap(xy2, out_xy, n * 2 * B) || (xy1 && overlap(xy1, out_xy, n * 2 * B)))   // see mul_var_dev
      return fail(E_ARG, "p1_xy / p2_xy and out_xy must not overlap");
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      const u8* p1 = xy1 ? xy1 + o * 2 * B : nullptr;
      if (custom_is_edwards(curve))
        rc = edc_chunk(1, m, k1 + o * B, p1, k2 + o * B, xy2 + o * 2 * B, nullptr, nullptr, out_xy + o * 2 * B,
                       out_inf + o);
      else if (is_custom(curve))
        rc = mul_add2_chunk<CvCustom>(m, k1 + o * B, p1, k2 + o * B, xy2 + o * 2 * B, out_xy + o * 2 * B,
                                      out_inf + o);
      else if (curve == CURVE_ED25519)
        rc = ed_mul_add2_chunk(m, k1 + o * B, p1, k2 + o * B, xy2 + o * 2 * B, out_xy + o * 2 * B,
                               out_inf + o);
      else
        ELL_SHORT_DISPATCH(curve, rc = p1 ? mul_add2_chunk<CV>(m, k1 + o * B, p1, k2 + o * B,
                                                               xy2 + o * 2 * B, out_xy + o * 2 * B,
                                                               out_inf + o)
                                              : mul_add_g_chunk<CV>(m, k1 + o * B, k2 + o * B,
                                                                    xy2 + o * 2 * B,
                                                                    out_xy + o * 2 * B, out_inf + o));
      if (rc) return rc;
    }
    return E_OK;
  }

  // ok: the verdicts, strictly 0 / 1.  st (may be null): the domain status per item -- 2 where the
  // key is not on the curve (verdict 0 there), else 0 (Work::store_verdict)
  int ecdsa_verify_dev(int curve, size_t n, const u8* hash, int hash_len, int msg_bits,
                       const u8* r, const u8* s, const u8* pub, u8* ok, u8* st) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve >= CURVE_ED25519)
      return fail(E_UNSUPPORTED, "ECDSA verify is implemented for the short Weierstrass presets");
    if (n && (!hash || !r || !s || !pub || !ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0 || msg_bits < 0) return fail(E_ARG, "bad hash_len / msg_bits");
    // _truncateToN (ec/index.js:97-102): delta = bitLength - n.bitLength()
    int bits = msg_bits ? msg_bits : hash_len * 8;
    int shift = bits - ci->order_bits;
    if (shift < 0) shift = 0;
    int ln = (ci->order_bits + 31) / 32;
    if (hash_len * 8 - shift > 32 * ln || hash_len - (shift >> 3) > 4 * (ln + 1))
      return fail(E_ARG, "hash_len / msg_bits combination leaves more bits than the order width");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    const size_t B = ci->field_bytes, NB = ci->order_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = ecdsa_chunk<CV>(m, hash + o * hash_len, hash_len, shift,
                                                     r + o * NB, s + o * NB, pub + o * 2 * B,
                                                     ok + o, st ? st + o : nullptr));
      if (rc) return rc;
    }
    return E_OK;
  }

  // out_bad (may be null): out_bad[i] = 1 where x[i] is no abscissa of the curve (KeyPair#derive's validate)
  int x25519_dev(size_t n, const u8* k, const u8* x, u8* out_x, u8* out_inf, u8* out_bad = nullptr) {
    if (n && (!k || !x || !out_x || !out_inf)) return fail(E_ARG, "null pointer");
    int rc = E_OK;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      rc = x25519_chunk(m, k + o * 32, x + o * 32, out_x + o * 32, out_inf + o, out_bad ? out_bad + o : nullptr);
      if (rc) return rc;
    }
    return E_OK;
  }

  // ECDSA sign with caller-supplied nonces (one pass of EC#sign's loop per item)
  int ecdsa_sign_dev(int curve, size_t n, const u8* hash, int hash_len, int msg_bits, const u8* priv,
                     const u8* nonces, int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve >= CURVE_ED25519)
      return fail(E_UNSUPPORTED, "ECDSA sign is implemented for the short Weierstrass presets");
    if (n && (!hash || !priv || !nonces || !out_r || !out_s || !out_recid || !out_ok))
      return fail(E_ARG, "null pointer");
    if (hash_len <= 0 || msg_bits < 0) return fail(E_ARG, "bad hash_len / msg_bits");
    int bits = msg_bits ? msg_bits : hash_len * 8;
    int shift = bits - ci->order_bits;
    if (shift < 0) shift = 0;
    int ln = (ci->order_bits + 31) / 32;
    if (hash_len * 8 - shift > 32 * ln || hash_len - (shift >> 3) > 4 * (ln + 1))
      return fail(E_ARG, "hash_len / msg_bits combination leaves more bits than the order width");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    const size_t NB = ci->order_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = sign_chunk<CV>(m, hash + o * hash_len, hash_len, shift, priv + o * NB,
                                                    nonces + o * NB, canonical, out_r + o * NB,
                                                    out_s + o * NB, out_recid + o, out_ok + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int ecdsa_sign_host(int curve, size_t n, const u8* hash, int hash_len, int msg_bits, const u8* priv,
                      const u8* nonces, int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!hash || !priv || !nonces || !out_r || !out_s || !out_recid || !out_ok))
      return fail(E_ARG, "null pointer");
    if (hash_len <= 0) return fail(E_ARG, "bad hash_len");
    size_t NB = ci->order_bytes;
    const size_t HL = (size_t)hash_len;
    u8* dh = out_buf(G_IN0, n * HL);
    u8* dd = out_buf(G_IN1, n * NB);
    u8* dk = out_buf(G_IN2, n * NB);
    u8* dr = out_buf(G_OUT0, n * NB * 2 + 2 * n);
    if (!dh || !dd || !dk || !dr) return fail(E_NOMEM, "staging allocation failed");
    u8* dsg = dr + n * NB;
    u8* drec = dsg + n * NB;
    u8* dok = drec + n;
    HostIn ins[3] = {{dh, hash, HL}, {dd, priv, NB}, {dk, nonces, NB}};
    HostOut outs[4] = {{out_r, dr, NB}, {out_s, dsg, NB}, {out_recid, drec, 1}, {out_ok, dok, 1}};
    return pipelined(n, ins, 3, outs, 4, [&](size_t o, size_t m) {
      return ecdsa_sign_dev(curve, m, dh + o * HL, hash_len, msg_bits, dd + o * NB, dk + o * NB, canonical,
                            dr + o * NB, dsg + o * NB, drec + o, dok + o);
    });
  }

  // EC#sign with the reference's own nonce source (HmacDRBG, deterministic): nonces == nullptr
  // in the calls below means "derive them"
  int ecdsa_sign_det_dev(int curve, size_t n, const u8* hash, int hash_len, int msg_bits, const u8* priv,
                         int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve >= CURVE_ED25519)
      return fail(E_UNSUPPORTED, "ECDSA sign is implemented for the short Weierstrass presets");
    if (n && (!hash || !priv || !out_r || !out_s || !out_recid || !out_ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0 || msg_bits < 0) return fail(E_ARG, "bad hash_len / msg_bits");
    int bits = msg_bits ? msg_bits : hash_len * 8;
    int shift = bits - ci->order_bits;
    if (shift < 0) shift = 0;
    int ln = (ci->order_bits + 31) / 32;
    if (hash_len * 8 - shift > 32 * ln || hash_len - (shift >> 3) > 4 * (ln + 1))
      return fail(E_ARG, "hash_len / msg_bits combination leaves more bits than the order width");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    const size_t NB = ci->order_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = sign_det_chunk<CV>(m, hash + o * hash_len, hash_len, shift, priv + o * NB,
                                                        canonical, out_r + o * NB, out_s + o * NB,
                                                        out_recid + o, out_ok + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int ecdsa_sign_det_host(int curve, size_t n, const u8* hash, int hash_len, int msg_bits, const u8* priv,
                          int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!hash || !priv || !out_r || !out_s || !out_recid || !out_ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0) return fail(E_ARG, "bad hash_len");
    const size_t NB = ci->order_bytes;
    const size_t HL = (size_t)hash_len;
    u8* dh = out_buf(G_IN0, n * HL);
    u8* dd = out_buf(G_IN1, n * NB);
    u8* dr = out_buf(G_OUT0, n * NB);
    u8* dsg = out_buf(G_OUT1, n * NB);
    u8* drec = out_buf(G_IN3, n);
    u8* dok = out_buf(G_IN4, n);
    if (!dh || !dd || !dr || !dsg || !drec || !dok) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dh, hash, HL}, {dd, priv, NB}};
    HostOut outs[4] = {{out_r, dr, NB}, {out_s, dsg, NB}, {out_recid, drec, 1}, {out_ok, dok, 1}};
    return pipelined(n, ins, 2, outs, 4, [&](size_t o, size_t m) {
      return ecdsa_sign_det_dev(curve, m, dh + o * HL, hash_len, msg_bits, dd + o * NB, canonical,
                                dr + o * NB, dsg + o * NB, drec + o, dok + o);
    });
  }

  // EC#recoverPubKey (ec/index.js:231-259) over a batch: status 0 point / 1 infinity /
  // 2 the reference throws / 3 outside the engine's domain (r = 0 or r >= n)
  int ecdsa_recover_dev(int curve, size_t n, const u8* hash, int hash_len, const u8* r, const u8* s,
                        const u8* recid, u8* out_xy, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve >= CURVE_ED25519)
      return fail(E_UNSUPPORTED, "public-key recovery is an ECDSA (short Weierstrass) operation");
    if (n && (!hash || !r || !s || !recid || !out_xy || !out_status)) return fail(E_ARG, "null pointer");
    int ln = (ci->order_bits + 31) / 32;
    if (hash_len <= 0 || hash_len > 8 * ln) return fail(E_ARG, "hash_len must be 1 .. twice the order width");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    const size_t B = ci->field_bytes, NB = ci->order_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = recover_chunk<CV>(m, hash + o * hash_len, hash_len, r + o * NB, s + o * NB,
                                                       recid + o, out_xy + o * 2 * B, out_status + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int ecdsa_recover_host(int curve, size_t n, const u8* hash, int hash_len, const u8* r, const u8* s,
                         const u8* recid, u8* out_xy, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!hash || !r || !s || !recid || !out_xy || !out_status)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0) return fail(E_ARG, "bad hash_len");
    const size_t B = ci->field_bytes, NB = ci->order_bytes;
    const size_t HL = (size_t)hash_len;
    u8* dh = out_buf(G_IN0, n * HL);
    u8* dr = out_buf(G_IN1, n * NB);
    u8* dsg = out_buf(G_IN2, n * NB);
    u8* dj = out_buf(G_IN3, n);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dst = out_buf(G_OUT1, n);
    if (!dh || !dr || !dsg || !dj || !dxy || !dst) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[4] = {{dh, hash, HL}, {dr, r, NB}, {dsg, s, NB}, {dj, recid, 1}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_status, dst, 1}};
    return pipelined(n, ins, 4, outs, 2, [&](size_t o, size_t m) {
      return ecdsa_recover_dev(curve, m, dh + o * HL, hash_len, dr + o * NB, dsg + o * NB, dj + o,
                               dxy + o * 2 * B, dst + o);
    });
  }

  // EdDSA (ed25519) verify.  msgs: concatenated message bytes; off (n+1 offsets, device
  // memory for the _dev form) or, if null, a uniform stride msg_len.
  int eddsa_verify_dev(size_t n, const u8* msgs, const u64* off, size_t msg_len, const u8* sigs,
                       const u8* pubs, u8* ok, u8* err) {
    if (n && (!sigs || !pubs || !ok || (!msgs && (off || msg_len)))) return fail(E_ARG, "null pointer");
    int rc = prepare_curve(CURVE_ED25519);
    if (rc) return rc;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      rc = eddsa_chunk(m, o, msgs, off, msg_len, sigs, pubs, ok, err);
      if (rc) return rc;
    }
    return E_OK;
  }
  int eddsa_verify_host(size_t n, const u8* msgs, const u64* off, size_t msg_len, const u8* sigs,
                        const u8* pubs, u8* ok, u8* err) {
    if (n && (!sigs || !pubs || !ok)) return fail(E_ARG, "null pointer");
    if (off)                       // len = off[i+1] - off[i] per item: offsets must not decrease
      for (size_t i = 0; i < n; i++)
        if (off[i] > off[i + 1]) return fail(E_ARG, "message offsets must be non-decreasing");
    size_t total = off ? (size_t)off[n] : n * msg_len;
    if (total && !msgs) return fail(E_ARG, "null message pointer");
    if (n && n <= bk.pipeline_quantum()) {
      // a few items: through the pinned buffer, one synchronisation (or none: ellgpu_ctx_defer)
      u8* dm = out_buf(G_IN0, total);
      u64* doff = off ? (u64*)out_buf(G_IN1, (n + 1) * sizeof(u64)) : nullptr;
      u8* dsg = out_buf(G_IN2, n * 64);
      u8* dpk = out_buf(G_IN3, n * 32);
      u8* dok = out_buf(G_OUT0, n);
      u8* derr = out_buf(G_OUT1, n);
      if (!dm || !dsg || !dpk || !dok || !derr || (off && !doff)) return fail(E_NOMEM, "staging allocation failed");
      SpanIn si[4] = {{dm, msgs ? msgs : (const u8*)"", total}, {(u8*)doff, (const u8*)off, off ? (n + 1) * sizeof(u64) : 0},
                      {dsg, sigs, n * 64}, {dpk, pubs, n * 32}};
      SpanOut so[2] = {{ok, dok, n}, {err, derr, n}};
      int rc = E_OK;
      if (small_call(si, 4, so, 2, [&]() { return eddsa_verify_dev(n, dm, doff, msg_len, dsg, dpk, dok, derr); }, &rc)) return rc;
    }
    defer_skip();
    u8* dm = put(G_IN0, msgs ? msgs : (const u8*)"", total);
    u64* doff = off ? (u64*)put(G_IN1, off, (n + 1) * sizeof(u64)) : nullptr;
    u8* dsg = put(G_IN2, sigs, n * 64);
    u8* dpk = put(G_IN3, pubs, n * 32);
    u8* dok = out_buf(G_OUT0, n);
    u8* derr = out_buf(G_OUT1, n);
    if (!dm || !dsg || !dpk || !dok || !derr || (off && !doff)) return fail(E_NOMEM, "staging allocation failed");
    int rc = eddsa_verify_dev(n, dm, doff, msg_len, dsg, dpk, dok, derr);
    if (rc) return rc;
    bk.d2h(ok, dok, n);
    if (err) bk.d2h(err, derr, n);
    return bk.sync();
  }

  // EdDSA (ed25519) sign from 32-byte secrets: EDDSA#sign with KeyPair.fromSecret.  Messages as
  // for eddsa_verify_dev; pub (n x 32, the encoded public keys) may be null.
  int eddsa_sign_dev(size_t n, const u8* secrets, const u8* msgs, const u64* off, size_t msg_len,
                     u8* sig, u8* pub) {
    if (n && (!secrets || !sig || (!msgs && (off || msg_len)))) return fail(E_ARG, "null pointer");
    int rc = prepare_curve(CURVE_ED25519);
    if (rc) return rc;
    const size_t step = CHUNK / 2;                 // two fixed-base multiplications per item
    for (size_t o = 0; o < n; o += step) {
      size_t m = n - o < step ? n - o : step;
      rc = eddsa_sign_chunk(m, o, secrets, msgs, off, msg_len, sig, pub);
      if (rc) return rc;
    }
    return E_OK;
  }
  int eddsa_sign_host(size_t n, const u8* secrets, const u8* msgs, const u64* off, size_t msg_len,
                      u8* sig, u8* pub) {
    if (n && (!secrets || !sig)) return fail(E_ARG, "null pointer");
    if (off)                       // len = off[i+1] - off[i] per item: offsets must not decrease
      for (size_t i = 0; i < n; i++)
        if (off[i] > off[i + 1]) return fail(E_ARG, "message offsets must be non-decreasing");
    size_t total = off ? (size_t)off[n] : n * msg_len;
    if (total && !msgs) return fail(E_ARG, "null message pointer");
    if (n && n <= bk.pipeline_quantum()) {
      u8* dm = out_buf(G_IN0, total);
      u64* doff = off ? (u64*)out_buf(G_IN1, (n + 1) * sizeof(u64)) : nullptr;
      u8* dsec = out_buf(G_IN2, n * 32);
      u8* dsig = out_buf(G_OUT0, n * 64);
      u8* dpub = out_buf(G_OUT1, n * 32);
      if (!dm || !dsec || !dsig || !dpub || (off && !doff)) return fail(E_NOMEM, "staging allocation failed");
      SpanIn si[3] = {{dm, msgs ? msgs : (const u8*)"", total}, {(u8*)doff, (const u8*)off, off ? (n + 1) * sizeof(u64) : 0},
                      {dsec, secrets, n * 32}};
      SpanOut so[2] = {{sig, dsig, n * 64}, {pub, dpub, n * 32}};
      int rc = E_OK;
      if (small_call(si, 3, so, 2, [&]() { return eddsa_sign_dev(n, dsec, dm, doff, msg_len, dsig, dpub); }, &rc)) return rc;
    }
    defer_skip();
    u8* dm = put(G_IN0, msgs ? msgs : (const u8*)"", total);
    u64* doff = off ? (u64*)put(G_IN1, off, (n + 1) * sizeof(u64)) : nullptr;
    u8* dsec = put(G_IN2, secrets, n * 32);
    u8* dsig = out_buf(G_OUT0, n * 64);
    u8* dpub = out_buf(G_OUT1, n * 32);
    if (!dm || !dsec || !dsig || !dpub || (off && !doff)) return fail(E_NOMEM, "staging allocation failed");
    int rc = eddsa_sign_dev(n, dsec, dm, doff, msg_len, dsig, dpub);
    if (rc) return rc;
    bk.d2h(sig, dsig, n * 64);
    if (pub) bk.d2h(pub, dpub, n * 32);
    return bk.sync();
  }

  // point decompression: short curves from x (pointFromX), ed25519 from y (pointFromY)
  int decompress_dev(int curve, size_t n, const u8* v, const u8* odd, u8* out_xy, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519) return fail(E_UNSUPPORTED, "curve25519 points are x-only");
    if (n && (!v || !odd || !out_xy || !out_ok)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    int rc = E_OK;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      if (curve == CURVE_ED25519) rc = ed_decompress_chunk(m, v + o * B, odd + o, out_xy + o * 2 * B, out_ok + o);
      else ELL_SHORT_DISPATCH(curve, rc = decompress_chunk<CV>(m, v + o * B, odd + o, out_xy + o * 2 * B, out_ok + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int decompress_host(int curve, size_t n, const u8* v, const u8* odd, u8* out_xy, u8* out_ok) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!v || !odd || !out_xy || !out_ok)) return fail(E_ARG, "null pointer");
    size_t B = ci->field_bytes;
    u8* dv = out_buf(G_IN0, n * B);
    u8* dodd = out_buf(G_IN1, n);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dok = out_buf(G_OUT1, n);
    if (!dv || !dodd || !dxy || !dok) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dv, v, B}, {dodd, odd, 1}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_ok, dok, 1}};
    return pipelined(n, ins, 2, outs, 2, [&](size_t o, size_t m) {
      return decompress_dev(curve, m, dv + o * B, dodd + o, dxy + o * 2 * B, dok + o);
    });
  }

  // ---- SEC1 / EdDSA point codecs and key validation --------------------------------
  enum { OP_DECODE = 0, OP_ENCODE = 1, OP_VALIDATE = 2 };
  // decodePoint (base.js:270-293; eddsa/index.js:99-109): n encodings of enc_len bytes each
  int decode_points_dev(int curve, size_t n, const u8* enc, size_t enc_len, u8* out_xy, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519) return fail(E_UNSUPPORTED, "curve25519 points are x-only");
    if (n && (!enc || !out_xy || !out_status)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    if (curve == CURVE_ED25519 && enc_len != 32) return fail(E_ARG, "ed25519 encodings are 32 bytes");
    if (enc_len == 0) return fail(E_ARG, "enc_len must be positive");
    int rc = E_OK;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      if (curve == CURVE_ED25519)
        rc = ed_codec_chunk(OP_DECODE, m, enc + o * enc_len, 0, nullptr, out_xy + o * 2 * B, out_status + o);
      else
        ELL_SHORT_DISPATCH(curve, rc = codec_chunk<CV>(OP_DECODE, m, enc + o * enc_len, enc_len, 0, nullptr,
                                                       out_xy + o * 2 * B, out_status + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  static size_t encoded_len(int curve, size_t B, int compact) {
    return curve == CURVE_ED25519 ? 32 : (compact ? 1 + B : 1 + 2 * B);
  }
  // BasePoint#encode (base.js:295-311) / EDDSA#encodePoint (eddsa/index.js:94-98)
  int encode_points_dev(int curve, size_t n, const u8* xy, int compact, u8* out_enc) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519) return fail(E_UNSUPPORTED, "curve25519 points are x-only");
    if (n && (!xy || !out_enc)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes, EL = encoded_len(curve, B, compact);
    int rc = E_OK;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      if (curve == CURVE_ED25519)
        rc = ed_codec_chunk(OP_ENCODE, m, xy + o * 2 * B, 0, nullptr, out_enc + o * EL, nullptr);
      else
        ELL_SHORT_DISPATCH(curve, rc = codec_chunk<CV>(OP_ENCODE, m, xy + o * 2 * B, 0, compact, nullptr,
                                                       out_enc + o * EL, nullptr));
      if (rc) return rc;
    }
    return E_OK;
  }
  // KeyPair#validate (ec/key.js:41-52): 0 ok, 1 'Invalid public key' (inf[i] set), 2 'Public key
  // is not a point', 3 'Public key * N != O' (only tested when check_order)
  int validate_dev(int curve, size_t n, const u8* xy, const u8* inf, int check_order, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519) return fail(E_UNSUPPORTED, "curve25519 points are x-only");
    if (n && (!xy || !out_status)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    int rc = E_OK;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      const u8* fi = inf ? inf + o : nullptr;
      if (curve == CURVE_ED25519)
        rc = ed_codec_chunk(OP_VALIDATE, m, xy + o * 2 * B, check_order, fi, nullptr, out_status + o);
      else
        ELL_SHORT_DISPATCH(curve, rc = codec_chunk<CV>(OP_VALIDATE, m, xy + o * 2 * B, 0, check_order, fi,
                                                       nullptr, out_status + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int decode_points_host(int curve, size_t n, const u8* enc, size_t enc_len, u8* out_xy, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!enc || !out_xy || !out_status)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    u8* de = out_buf(G_IN0, n * enc_len);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dst = out_buf(G_OUT1, n);
    if (!de || !dxy || !dst) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[1] = {{de, enc, enc_len}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_status, dst, 1}};
    return pipelined(n, ins, 1, outs, 2, [&](size_t o, size_t m) {
      return decode_points_dev(curve, m, de + o * enc_len, enc_len, dxy + o * 2 * B, dst + o);
    });
  }
  int encode_points_host(int curve, size_t n, const u8* xy, int compact, u8* out_enc) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!xy || !out_enc)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes, EL = encoded_len(curve, B, compact);
    u8* dxy = out_buf(G_IN0, n * 2 * B);
    u8* de = out_buf(G_OUT0, n * EL);
    if (!dxy || !de) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[1] = {{dxy, xy, 2 * B}};
    HostOut outs[1] = {{out_enc, de, EL}};
    return pipelined(n, ins, 1, outs, 1, [&](size_t o, size_t m) {
      return encode_points_dev(curve, m, dxy + o * 2 * B, compact, de + o * EL);
    });
  }
  int validate_host(int curve, size_t n, const u8* xy, const u8* inf, int check_order, u8* out_status) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!xy || !out_status)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    u8* dxy = out_buf(G_IN0, n * 2 * B);
    u8* dinf = inf ? out_buf(G_IN1, n) : nullptr;
    u8* dst = out_buf(G_OUT0, n);
    if (!dxy || !dst || (inf && !dinf)) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dxy, xy, 2 * B}, {dinf, inf, 1}};
    HostOut outs[1] = {{out_status, dst, 1}};
    return pipelined(n, ins, inf ? 2 : 1, outs, 1, [&](size_t o, size_t m) {
      return validate_dev(curve, m, dxy + o * 2 * B, dinf ? dinf + o : nullptr, check_order, dst + o);
    });
  }

  // ---- Point#add on affine points (short.js:365-412, edwards.js:350-360) ----------------
  int point_add_dev(int curve, size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2,
                    u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve == CURVE_CURVE25519) return fail(E_UNSUPPORTED, "Not supported on Montgomery curve");   // mont.js:103-105
    if (n && (!xy1 || !xy2 || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    int rc = E_OK;
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      const u8* i1 = inf1 ? inf1 + o : nullptr;
      const u8* i2 = inf2 ? inf2 + o : nullptr;
      if (custom_is_edwards(curve))
        rc = edc_chunk(2, m, nullptr, xy1 + o * 2 * B, nullptr, xy2 + o * 2 * B, i1, i2, out_xy + o * 2 * B, out_inf + o);
      else if (is_custom(curve))
        rc = point_add_chunk<CvCustom>(m, xy1 + o * 2 * B, i1, xy2 + o * 2 * B, i2, out_xy + o * 2 * B, out_inf + o);
      else if (curve == CURVE_ED25519)
        rc = ed_point_add_chunk(m, xy1 + o * 2 * B, i1, xy2 + o * 2 * B, i2, out_xy + o * 2 * B, out_inf + o);
      else
        ELL_SHORT_DISPATCH(curve, rc = point_add_chunk<CV>(m, xy1 + o * 2 * B, i1, xy2 + o * 2 * B, i2,
                                                           out_xy + o * 2 * B, out_inf + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  int point_add_host(int curve, size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2,
                     u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!xy1 || !xy2 || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    const size_t B = ci->field_bytes;
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    u8* d1 = out_buf(G_IN0, n * 2 * B);
    u8* d2 = out_buf(G_IN1, n * 2 * B);
    u8* di1 = inf1 ? out_buf(G_IN2, n) : nullptr;
    u8* di2 = inf2 ? out_buf(G_IN3, n) : nullptr;
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dinf = out_buf(G_OUT1, n);
    if (!d1 || !d2 || !dxy || !dinf || (inf1 && !di1) || (inf2 && !di2)) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[4] = {{d1, xy1, 2 * B}, {d2, xy2, 2 * B}, {nullptr, nullptr, 0}, {nullptr, nullptr, 0}};
    int ni = 2;
    if (inf1) ins[ni++] = HostIn{di1, inf1, 1};
    if (inf2) ins[ni++] = HostIn{di2, inf2, 1};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_inf, dinf, 1}};
    return pipelined(n, ins, ni, outs, 2, [&](size_t o, size_t m) {
      return point_add_dev(curve, m, d1 + o * 2 * B, di1 ? di1 + o : nullptr, d2 + o * 2 * B,
                           di2 ? di2 + o : nullptr, dxy + o * 2 * B, dinf + o);
    });
  }

  // ---- signature DER codec and EC#verify on wire formats ------------------------------
  enum { OP_FROM_DER = 0, OP_TO_DER = 1, OP_WIRE_STATUS = 2 };
  int check_short(int curve, const CurveInfo*& ci) {
    ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (curve >= CURVE_ED25519) return fail(E_UNSUPPORTED, "ECDSA signatures belong to the short Weierstrass presets");
    return E_OK;
  }
  // Signature#_importDER (ec/signature.js:83-147): der = n records of `stride` bytes, der_len[i] used
  int sig_from_der_dev(int curve, size_t n, const u8* der, size_t stride, const u32* der_len,
                       u8* out_r, u8* out_s, u8* out_status) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!der || !der_len || !out_r || !out_s || !out_status)) return fail(E_ARG, "null pointer");
    if (stride == 0) return fail(E_ARG, "stride must be positive");
    const size_t NB = ci->order_bytes;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = der_chunk<CV>(OP_FROM_DER, m, der + o * stride, nullptr, stride,
                                                   (u32*)der_len + o, out_r + o * NB, out_s + o * NB,
                                                   out_status + o));
      if (rc) return rc;
    }
    return E_OK;
  }
  // Signature#toDER (ec/signature.js:149-176): out = n records of `stride` >= 2 NB + 9 bytes
  int sig_to_der_dev(int curve, size_t n, const u8* r, const u8* s, u8* out_der, size_t stride, u32* out_len) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!r || !s || !out_der || !out_len)) return fail(E_ARG, "null pointer");
    const size_t NB = ci->order_bytes;
    if (stride < 2 * NB + 9) return fail(E_ARG, "stride must be at least 2 * order_bytes + 9");
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      ELL_SHORT_DISPATCH(curve, rc = der_chunk<CV>(OP_TO_DER, m, r + o * NB, s + o * NB, stride, out_len + o,
                                                   out_der + o * stride, nullptr, nullptr));
      if (rc) return rc;
    }
    return E_OK;
  }
  // EC#verify(msg, derSignature, encodedKey) (ec/index.js:188-229 with keyFromPublic ->
  // decodePoint and new Signature(der)): decode, parse and verify on the device
  int ecdsa_verify_wire_dev(int curve, size_t n, const u8* hash, int hash_len, int msg_bits,
                            const u8* der, size_t der_stride, const u32* der_len, const u8* pub_enc,
                            size_t pub_len, u8* ok, u8* err) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!hash || !der || !der_len || !pub_enc || !ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0 || der_stride == 0 || pub_len == 0) return fail(E_ARG, "bad hash_len / stride / pub_len");
    const size_t B = ci->field_bytes, NB = ci->order_bytes, HL = (size_t)hash_len;
    for (size_t o = 0; o < n; o += CHUNK) {
      size_t m = n - o < CHUNK ? n - o : CHUNK;
      u8* tmp = (u8*)scratch(S_WIRE, m * (2 * NB + 2 * B + 3));
      if (!tmp) return fail(E_NOMEM, "scratch allocation failed");
      u8* r = tmp;
      u8* s = r + m * NB;
      u8* xy = s + m * NB;
      u8* kst = xy + m * 2 * B;
      u8* sst = kst + m;
      u8* vst = sst + m;
      rc = decode_points_dev(curve, m, pub_enc + o * pub_len, pub_len, xy, kst);
      if (rc) return rc;
      rc = sig_from_der_dev(curve, m, der + o * der_stride, der_stride, der_len + o, r, s, sst);
      if (rc) return rc;
      rc = ecdsa_verify_dev(curve, m, hash + o * HL, hash_len, msg_bits, r, s, xy, ok + o, vst);
      if (rc) return rc;
      ELL_SHORT_DISPATCH(curve, rc = der_chunk<CV>(OP_WIRE_STATUS, m, kst, sst, 0, nullptr, ok + o,
                                                   err ? err + o : nullptr, vst));
      if (rc) return rc;
    }
    return E_OK;
  }
  int sig_from_der_host(int curve, size_t n, const u8* der, size_t stride, const u32* der_len,
                        u8* out_r, u8* out_s, u8* out_status) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!der || !der_len || !out_r || !out_s || !out_status)) return fail(E_ARG, "null pointer");
    const size_t NB = ci->order_bytes;
    u8* dd = out_buf(G_IN0, n * stride);
    u8* dl = out_buf(G_IN1, n * 4);
    u8* dr = out_buf(G_OUT0, n * NB);
    u8* ds = out_buf(G_OUT1, n * NB);
    u8* dst = out_buf(G_IN2, n);
    if (!dd || !dl || !dr || !ds || !dst) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dd, der, stride}, {dl, (const u8*)der_len, 4}};
    HostOut outs[3] = {{out_r, dr, NB}, {out_s, ds, NB}, {out_status, dst, 1}};
    return pipelined(n, ins, 2, outs, 3, [&](size_t o, size_t m) {
      return sig_from_der_dev(curve, m, dd + o * stride, stride, (const u32*)dl + o, dr + o * NB, ds + o * NB, dst + o);
    });
  }
  int sig_to_der_host(int curve, size_t n, const u8* r, const u8* s, u8* out_der, size_t stride, u32* out_len) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!r || !s || !out_der || !out_len)) return fail(E_ARG, "null pointer");
    const size_t NB = ci->order_bytes;
    u8* dr = out_buf(G_IN0, n * NB);
    u8* ds = out_buf(G_IN1, n * NB);
    u8* dd = out_buf(G_OUT0, n * stride);
    u8* dl = out_buf(G_OUT1, n * 4);
    if (!dr || !ds || !dd || !dl) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dr, r, NB}, {ds, s, NB}};
    HostOut outs[2] = {{out_der, dd, stride}, {(u8*)out_len, dl, 4}};
    return pipelined(n, ins, 2, outs, 2, [&](size_t o, size_t m) {
      return sig_to_der_dev(curve, m, dr + o * NB, ds + o * NB, dd + o * stride, stride, (u32*)dl + o);
    });
  }
  int ecdsa_verify_wire_host(int curve, size_t n, const u8* hash, int hash_len, int msg_bits,
                             const u8* der, size_t der_stride, const u32* der_len, const u8* pub_enc,
                             size_t pub_len, u8* ok, u8* err) {
    const CurveInfo* ci;
    int rc = check_short(curve, ci);
    if (rc) return rc;
    if (n && (!hash || !der || !der_len || !pub_enc || !ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0) return fail(E_ARG, "bad hash_len");
    const size_t HL = (size_t)hash_len;
    u8* dh = out_buf(G_IN0, n * HL);
    u8* dd = out_buf(G_IN1, n * der_stride);
    u8* dl = out_buf(G_IN2, n * 4);
    u8* dq = out_buf(G_IN3, n * pub_len);
    u8* dok = out_buf(G_OUT0, n);
    u8* derr = out_buf(G_OUT1, n);
    if (!dh || !dd || !dl || !dq || !dok || !derr) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[4] = {{dh, hash, HL}, {dd, der, der_stride}, {dl, (const u8*)der_len, 4}, {dq, pub_enc, pub_len}};
    HostOut outs[2] = {{ok, dok, 1}, {err, derr, 1}};
    return pipelined(n, ins, 4, outs, err ? 2 : 1, [&](size_t o, size_t m) {
      return ecdsa_verify_wire_dev(curve, m, dh + o * HL, hash_len, msg_bits, dd + o * der_stride, der_stride,
                                   (const u32*)dl + o, dq + o * pub_len, pub_len, dok + o, derr + o);
    });
  }

  // ---- host-buffer wrappers: stage through device buffers --------------------
  u8* put(int slot, const void* host, size_t bytes) {
    if (!host) return nullptr;
    u8* d = (u8*)staging(slot, bytes ? bytes : 1);
    if (d && bytes) bk.h2d(d, host, bytes);
    return d;
  }
  u8* out_buf(int slot, size_t bytes) { return (u8*)staging(slot, bytes ? bytes : 1); }

  // Host-buffer calls are software-pipelined: the batch is cut into chunks that alternate
  // between two compute lanes (stream + scratch arena each), so chunk c+1's wavefronts fill
  // the SIMDs as chunk c's drain (a lone chunk pays ~1 ms of ramp-down at 1 wave round), and
  // chunk c+1's H2D / chunk c-1's D2H run on the copy stream while chunk c computes.  The
  // first chunk is one residency quantum so that compute starts early.  body(o, m) launches
  // items [o, o+m) on the current lane.
  struct HostIn { u8* dev; const u8* host; size_t stride; };
  struct HostOut { u8* host; const u8* dev; size_t stride; };
  // A batch of a few items -- one patched EC#verify, Point#mul or EC#sign -- has nothing to
  // pipeline: its inputs and results travel through ONE pinned host buffer, on the context's own
  // stream, with one synchronisation (no copy streams, no events, no pageable-memory copies that
  // each block the host: 60 -> ~25 us of overhead per call).
  static constexpr size_t SMALL_HOST_BYTES = 256 * 1024;
  // The SPLIT form of such a call (ellgpu_ctx_defer / ellgpu_ctx_collect): an armed context's next
  // small call returns as soon as its copies and kernels are enqueued; the synchronisation and the
  // copy into the caller's result buffers happen in defer_collect().  The host thread is free in
  // between -- the JavaScript layer re-validates what it took on trust (the reference's
  // precomputed tables, elliptic_amd/js/index.js) while the device works.  Anything else that
  // enters the context first completes the pending call (capi_common.h ELL_LOCK).
  struct Deferred {
    bool armed = false, on = false;
    std::thread::id owner;         // the thread that armed: only ITS next call is deferred (a libuv worker's
                                   // batch that slips in between must run to completion as usual)
    struct Out { u8* host; size_t bytes; } outs[4];
    int nout = 0;
    size_t out0 = 0;
    u8* pin = nullptr;
  } defer_;
  void defer_arm() { defer_.armed = true; defer_.owner = std::this_thread::get_id(); }
  bool defer_pending() const { return defer_.on; }
  // is the next small call of THIS thread to be deferred?  (consumes the arming if so)
  bool defer_take() {
    if (!defer_.armed || defer_.owner != std::this_thread::get_id()) return false;
    defer_skip();
    return true;
  }
  // a call of the arming thread that cannot be deferred disarms; another thread's call leaves it armed
  void defer_skip() {
    if (defer_.armed && defer_.owner == std::this_thread::get_id()) defer_.armed = false;
  }
  static size_t pad64(size_t b) { return (b + 63) & ~(size_t)63; }
  int defer_collect() {
    defer_skip();
    if (!defer_.on) return E_OK;
    defer_.on = false;
    bk.select_lane(0);
    const int rs = bk.sync();
    size_t off = defer_.out0;
    for (int i = 0; i < defer_.nout; i++)
      if (defer_.outs[i].host) {
        if (!rs) memcpy(defer_.outs[i].host, defer_.pin + off, defer_.outs[i].bytes);
        off += pad64(defer_.outs[i].bytes);
      }
    if (rs) fail(rs, "device error in a deferred call");
    return rs;
  }
  // the few-item call itself: operands (host -> pinned -> device), body, results (device -> pinned
  // [-> host, now or in defer_collect]).  Returns false when the call does not fit the pinned buffer.
  struct SpanIn { u8* dev; const u8* host; size_t bytes; };
  struct SpanOut { u8* host; const u8* dev; size_t bytes; };
  template <class Body>
  bool small_call(const SpanIn* ins, int nin, const SpanOut* outs, int nout, Body body, int* result) {
    size_t tot = 0;
    for (int i = 0; i < nin; i++) if (ins[i].host) tot += pad64(ins[i].bytes);
    for (int i = 0; i < nout; i++) if (outs[i].host) tot += pad64(outs[i].bytes);
    u8* pin = tot <= SMALL_HOST_BYTES ? (u8*)bk.pinned(SMALL_HOST_BYTES) : nullptr;
    if (!pin) return false;
    const bool defer = defer_take();
    size_t off = 0;
    for (int i = 0; i < nin; i++)
      if (ins[i].host) {
        memcpy(pin + off, ins[i].host, ins[i].bytes);
        if (ins[i].bytes) bk.h2d(ins[i].dev, pin + off, ins[i].bytes);
        off += pad64(ins[i].bytes);
      }
    lane_ = 0;
    const int rc = body();
    const size_t out0 = off;
    for (int i = 0; i < nout; i++)
      if (outs[i].host) {
        if (outs[i].bytes) bk.d2h(pin + off, outs[i].dev, outs[i].bytes);
        off += pad64(outs[i].bytes);
      }
    if (defer && !rc && nout <= 4 && !custom_active_) {
      defer_.on = true;
      defer_.nout = nout;
      for (int i = 0; i < nout; i++) defer_.outs[i] = {outs[i].host, outs[i].bytes};
      defer_.out0 = out0;
      defer_.pin = pin;
      *result = E_OK;
      return true;
    }
    const int rs = bk.sync();
    off = out0;
    for (int i = 0; i < nout; i++)
      if (outs[i].host) {
        if (!rc && !rs) memcpy(outs[i].host, pin + off, outs[i].bytes);
        off += pad64(outs[i].bytes);
      }
    *result = rc ? rc : rs;
    return true;
  }
  template <class Body>
  int pipelined(size_t n, const HostIn* ins, int nin, const HostOut* outs, int nout, Body body) {
    if (n && n <= bk.pipeline_quantum() && nin <= 4 && nout <= 4) {
      SpanIn si[4];
      SpanOut so[4];
      for (int i = 0; i < nin; i++) si[i] = {ins[i].dev, ins[i].host, n * ins[i].stride};
      for (int i = 0; i < nout; i++) so[i] = {outs[i].host, outs[i].dev, n * outs[i].stride};
      int rc = E_OK;
      if (small_call(si, nin, so, nout, [&]() { return body(0, n); }, &rc)) return rc;
    }
    defer_skip();
    size_t q = bk.pipeline_quantum(), step = q;
    size_t o = 0, po = 0, pm = 0;
    int pev = -1, rc = E_OK;
    for (int c = 0; o < n && !rc; c++) {
      size_t m = n - o;
      if (m > step + step / 2) m = step;          // a short tail is absorbed by the last chunk
      for (int i = 0; i < nin; i++)
        if (ins[i].host)
          bk.h2d_copy(ins[i].dev + o * ins[i].stride, ins[i].host + o * ins[i].stride, m * ins[i].stride);
      lane_ = c & 1;
      bk.select_lane(lane_);
      bk.copies_before_compute();
      rc = body(o, m);
      int ev = bk.mark_compute();
      if (pev >= 0) {
        bk.copy_after(pev);
        for (int i = 0; i < nout; i++)
          bk.d2h_copy(outs[i].host + po * outs[i].stride, outs[i].dev + po * outs[i].stride, pm * outs[i].stride);
      }
      pev = ev; po = o; pm = m;
      o += m;
      step = pipe_step_ * q;                      // measured: every chunk boundary costs ~0.4 ms
    }
    if (pev >= 0 && !rc) {
      bk.copy_after(pev);
      for (int i = 0; i < nout; i++)
        bk.d2h_copy(outs[i].host + po * outs[i].stride, outs[i].dev + po * outs[i].stride, pm * outs[i].stride);
    }
    lane_ = 0;
    bk.select_lane(0);
    int rs = bk.sync_lanes();
    return rc ? rc : rs;
  }

  int mul_fixed_host(int curve, size_t n, const u8* k, u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!k || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    size_t B = ci->field_bytes;
    u8* dk = out_buf(G_IN0, n * B);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dinf = out_buf(G_OUT1, n);
    if (!dk || !dxy || !dinf) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[1] = {{dk, k, B}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_inf, dinf, 1}};
    return pipelined(n, ins, 1, outs, 2, [&](size_t o, size_t m) {
      return mul_fixed_dev(curve, m, dk + o * B, dxy + o * 2 * B, dinf + o);
    });
  }
  int mul_var_host(int curve, size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!k || !xy || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    size_t B = ci->field_bytes;
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    u8* dk = out_buf(G_IN0, n * B);
    u8* dp = out_buf(G_IN1, n * 2 * B);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dinf = out_buf(G_OUT1, n);
    if (!dk || !dp || !dxy || !dinf) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dk, k, B}, {dp, xy, 2 * B}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_inf, dinf, 1}};
    return pipelined(n, ins, 2, outs, 2, [&](size_t o, size_t m) {
      return mul_var_dev(curve, m, dk + o * B, dp + o * 2 * B, dxy + o * 2 * B, dinf + o);
    });
  }
  int mul_add2_host(int curve, size_t n, const u8* k1, const u8* xy1, const u8* k2,
                    const u8* xy2, u8* out_xy, u8* out_inf) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!k1 || !k2 || !xy2 || !out_xy || !out_inf)) return fail(E_ARG, "null pointer");
    size_t B = ci->field_bytes;
    CustomScope sc(this, curve);
    if (sc.rc) return sc.rc;
    u8* d1 = out_buf(G_IN0, n * B);
    u8* dp1 = xy1 ? out_buf(G_IN1, n * 2 * B) : nullptr;
    u8* d2 = out_buf(G_IN2, n * B);
    u8* dp2 = out_buf(G_IN3, n * 2 * B);
    u8* dxy = out_buf(G_OUT0, n * 2 * B);
    u8* dinf = out_buf(G_OUT1, n);
    if (!d1 || !d2 || !dp2 || !dxy || !dinf || (xy1 && !dp1))
      return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[4] = {{d1, k1, B}, {dp1, xy1, 2 * B}, {d2, k2, B}, {dp2, xy2, 2 * B}};
    HostOut outs[2] = {{out_xy, dxy, 2 * B}, {out_inf, dinf, 1}};
    return pipelined(n, ins, 4, outs, 2, [&](size_t o, size_t m) {
      return mul_add2_dev(curve, m, d1 + o * B, dp1 ? dp1 + o * 2 * B : nullptr, d2 + o * B,
                          dp2 + o * 2 * B, dxy + o * 2 * B, dinf + o);
    });
  }
  int ecdsa_verify_host(int curve, size_t n, const u8* hash, int hash_len, int msg_bits,
                        const u8* r, const u8* s, const u8* pub, u8* ok, u8* st) {
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    if (n && (!hash || !r || !s || !pub || !ok)) return fail(E_ARG, "null pointer");
    if (hash_len <= 0) return fail(E_ARG, "bad hash_len");
    size_t B = ci->field_bytes, NB = ci->order_bytes;
    size_t HL = (size_t)hash_len;
    u8* dh = out_buf(G_IN0, n * HL);
    u8* dr = out_buf(G_IN1, n * NB);
    u8* dsg = out_buf(G_IN2, n * NB);
    u8* dq = out_buf(G_IN3, n * 2 * B);
    u8* dok = out_buf(G_OUT0, n);
    u8* dst = st ? out_buf(G_OUT1, n) : nullptr;
    if (!dh || !dr || !dsg || !dq || !dok || (st && !dst)) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[4] = {{dh, hash, HL}, {dr, r, NB}, {dsg, s, NB}, {dq, pub, 2 * B}};
    HostOut outs[2] = {{ok, dok, 1}, {st, dst, 1}};
    return pipelined(n, ins, 4, outs, st ? 2 : 1, [&](size_t o, size_t m) {
      return ecdsa_verify_dev(curve, m, dh + o * HL, hash_len, msg_bits, dr + o * NB, dsg + o * NB,
                              dq + o * 2 * B, dok + o, dst ? dst + o : nullptr);
    });
  }
  int x25519_host(size_t n, const u8* k, const u8* x, u8* out_x, u8* out_inf, u8* out_bad = nullptr) {
    if (n && (!k || !x || !out_x || !out_inf)) return fail(E_ARG, "null pointer");
    u8* dk = out_buf(G_IN0, n * 32);
    u8* dx = out_buf(G_IN1, n * 32);
    u8* dox = out_buf(G_OUT0, n * 32);
    u8* dinf = out_buf(G_OUT1, n);
    u8* dbad = out_bad ? out_buf(G_OUT2, n) : nullptr;
    if (!dk || !dx || !dox || !dinf || (out_bad && !dbad)) return fail(E_NOMEM, "staging allocation failed");
    HostIn ins[2] = {{dk, k, 32}, {dx, x, 32}};
    HostOut outs[3] = {{out_x, dox, 32}, {out_inf, dinf, 1}, {out_bad, dbad, 1}};
    return pipelined(n, ins, 2, outs, out_bad ? 3 : 2, [&](size_t o, size_t m) {
      return x25519_dev(m, dk + o * 32, dx + o * 32, dox + o * 32, dinf + o, dbad ? dbad + o : nullptr);
    });
  }

  // scratch arena of the current call (0 / 1: the compute lanes of pipelined(); a *_dev call takes
  // the lane of its stream, HipBackend::use_stream_dev)
  void set_lane(int l) { lane_ = l & 1; }
  int reserve(int curve, size_t n) {
    // run a dummy-sized allocation pass by touching the arena the way a verify /
    // mul_add2 batch of n items would
    const CurveInfo* ci = curve_info(curve);
    if (!ci) return fail(E_ARG, "unknown curve id");
    int rc = prepare_curve(curve);
    if (rc) return rc;
    size_t m = n < CHUNK ? n : CHUNK;
    size_t L = (size_t)(ci->field_bytes + 3) / 4 + 1;       // +1: the 29-bit secp256k1 field stores 9 limbs
    size_t ent = (curve == CURVE_ED25519) ? 16 * 4 * L * 4 : 32 * 3 * L * 4;
    // both lanes: a caller that alternates two streams works in both arenas
    for (int l = 0; l < 2; l++) {
      lane_ = l;
      if (!scratch(S_TBL, m * ent) || !scratch(S_JAC, m * 4 * L * 4) || !scratch(S_PRE, m * L * 4) ||
          !scratch(S_U12, m * 2 * L * 4) || !scratch(S_VALID, m)) {
        lane_ = 0;
        return fail(E_NOMEM, "scratch allocation failed");
      }
    }
    lane_ = 0;
    return E_OK;
  }

 private:
  void* comb_[CURVE_COUNT];                  // entry 0 of the fixed-base table (the slot in front of it: its geometry)
  void* comb_base_[CURVE_COUNT];             // what was allocated
  int comb_bits_[CURVE_COUNT];               // window width in use (signed combs may be narrower than the default)
  Buf scratch_[2][S_COUNT];   // one scratch arena per compute lane (see pipelined())
  int lane_ = 0;
  Buf staging_[G_COUNT];
  std::vector<RtField> custom_;  // user-defined curves of this context (id = CURVE_CUSTOM0 + index)
  bool custom_active_ = false;
  size_t pipe_step_ = pipe_step_default();   // chunks after the first, in quanta
};

// ---- out-of-class definitions of the per-(curve, operation) members: NOT inline, so that
// `extern template` (engine_extern.h) really keeps their kernels out of other TUs ----
template <class BK>
template <class CV>
int Engine<BK>::ensure_comb() {
  typedef Work<CV> W;
  if (comb_[CV::ID]) return E_OK;
  // Built by the engine itself: the variable-base kernel on the scalars d << (COMB_BITS w) and
  // the generator, in slices of at most 2^20 entries (the 22-bit signed comb of the 256-bit
  // curves has 25 M entries = 1.6 GB; a slice needs 1 GB of window-table scratch).
  // A SIGNED comb (the 256-bit curves) carries its window width in the slot in front of entry 0
  // (ladder.h comb_bits_of): when the device cannot hold the default table -- 12 windows x 2^21
  // entries = 1.6 GB per curve -- the SAME kernels run on a narrower one: 16 bits (17 windows,
  // 36 MB), 12, 8, 4.  ELLGPU_COMB_MAX_BYTES (developer / test override, read when the context is
  // created) refuses larger tables as if the allocation had failed.
  const int B = W::BYTES;
  int cb = W::COMB_BITS;
  for (;;) {
    const size_t n = W::COMB_SIGNED ? (size_t)comb_windows(8 * B, cb) << (cb - 1) : W::COMB_ENTRIES;
    const size_t bytes = (n + 1) * sizeof(typename W::A);
    const size_t slice = n < (size_t)ELL_COMB_SLICE ? n : (size_t)ELL_COMB_SLICE;
    typename W::A* base = (tune_.comb_max_bytes && bytes > tune_.comb_max_bytes) ? nullptr : (typename W::A*)bk.alloc(bytes);
    u8* dk = base ? (u8*)bk.alloc(slice * B) : nullptr;
    u8* dp = dk ? (u8*)bk.alloc(slice * 2 * B) : nullptr;
    int rc = (base && dk && dp) ? E_OK : E_NOMEM;
    if (rc == E_OK) {
      typename W::A* comb = base + 1;
      u32 hdr[sizeof(typename W::A) / 4] = {0};
      hdr[0] = (u32)cb;
      bk.h2d(base, hdr, sizeof hdr);
      for (size_t first = 0; first < n && rc == E_OK; first += slice) {
        const size_t m = n - first < slice ? n - first : slice;
        FnCombGen<CV> g{m, first, dk, dp, cb};
        bk.launch(g, m);
        rc = mul_var_chunk<CV>(m, dk, dp, nullptr, nullptr, comb + first);
      }
      const int src = bk.sync();                   // a failed launch (comb_gen included) surfaces here
      if (rc == E_OK && src != E_OK) rc = fail(src, "building the fixed-base table failed on the device");
      if (rc == E_OK) {
        bk.free_(dk);
        bk.free_(dp);
        comb_[CV::ID] = comb;
        comb_base_[CV::ID] = base;
        comb_bits_[CV::ID] = cb;
        return E_OK;
      }
    }
    if (dp) bk.free_(dp);
    if (dk) bk.free_(dk);
    if (base) bk.free_(base);
    // out of memory (the table, or the window-table scratch of its build): a narrower comb
    if (rc != E_NOMEM || !W::COMB_SIGNED || cb <= 4) return rc == E_NOMEM ? fail(E_NOMEM, "comb table allocation failed") : rc;
    err.clear();
    cb = cb > 16 ? 16 : (cb > 12 ? 12 : (cb > 8 ? 8 : 4));
  }
}


template <class BK>
template <class CV>
int Engine<BK>::normalize_chunk(size_t n, const u32* jac, u8* out_xy, u8* out_inf, typename Work<CV>::A* raw) {
  typedef Work<CV> W;
  u32* pre = (u32*)scratch(S_PRE, n * W::NS * 4);
  if (!pre) return fail(E_NOMEM, "scratch allocation failed");
  const int K = norm_batch_for(n);
  size_t T = (n + K - 1) / K;
  FnNormalize<CV> f{T, n, K, jac, pre, out_xy, out_inf, raw};
  bk.launch(f, T);
  return E_OK;
}


template <class BK>
template <class CV>
int Engine<BK>::mul_var_chunk(size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf,
                  typename Work<CV>::A* raw) {
  typedef Work<CV> W;
  // the window-table scratch is sized by the tuning that is launched (the small-grid tuning's
  // 5-bit windows take twice the slots per item of the full-grid tuning's)
  bool wide = false;
  if constexpr (CV::ENDO && W::L <= 8) wide = n <= small_grid();
  const size_t slots = wide ? (size_t)W::template stride<true>() : (size_t)W::template stride<false>();
  // (the parted form: a table and a result per half)
  const bool parted = wide && n <= parted_grid();
  typename W::VT* tbl = (typename W::VT*)scratch(S_TBL, (parted ? 2 : 1) * n * slots * sizeof(typename W::VT));
  u32* jac = (u32*)scratch(S_JAC, (parted ? 2 : 1) * n * 3 * W::NS * 4);
  if (!tbl || !jac) return fail(E_NOMEM, "scratch allocation failed");
  bool launched = false;
  if constexpr (CV::ENDO && W::L <= 8) {
    if (parted) {                           // most SIMDs would idle: two lanes per item, then the join
      if (!rows_for(n) && n <= coop_grid()) {   // ... or two WAVES per item (coop.h)
        FnMulPartsC fc{n, k, xy, jac, nullptr, nullptr};
        bk.launch_coop(fc, 2 * n);
      } else if (rows_for(n)) {             // ... or one ROW per item and half: four items per wave
        FnMulPartsR fr{n, k, xy, jac, nullptr, nullptr};
        bk.launch_coop(fr, 2 * ((n + 3) / 4));
      } else {
      const size_t npad = (n + 127) & ~(size_t)127;            // whole workgroups per half
      FnMulParts<CV> fp{n, npad, k, xy, tbl, jac, nullptr, nullptr};
      launch_fn(fp, npad + n);
      }
      FnMulJoin<CV> fj{n, jac, false, xy, out_xy, out_inf, raw};   // ... to affine, and the domain test
      return launch_fn(fj, n);
    } else if (wide) {                      // at most three waves per SIMD: the register-rich tuning
      FnMulVar<CV, 3, true> f{n, k, xy, tbl, jac};
      launch_fn(f, n);
      launched = true;
    }
  }
  if constexpr (!CV::ENDO && CoopNist<CV>::AVAILABLE) {
    // a handful of items: the ladder of every item on a wave of its own (the row layer); the
    // normalisation and the domain test below are the one-lane kernels'
    if (!launched && n <= coop_grid_of<CV>() && out_inf) {
      FnMulPartsN<CV> fc{n, k, xy, jac, nullptr, nullptr};
      bk.launch_coop(fc, n);
      launched = true;
    }
  }
  if (launched) {
  } else if (W::L > 12 && n > ELL_P521_PAIR_MIN) {
    FnMulVar<CV, (W::L > 12 ? 2 : 0)> f{n, k, xy, tbl, jac};
    bk.launch(f, n);
  } else {
    FnMulVar<CV> f{n, k, xy, tbl, jac};
    bk.launch(f, n);
  }
  int rc = normalize_chunk<CV>(n, jac, out_xy, out_inf, raw);
  if (rc == E_OK && out_inf) {                  // (the comb build has no out_inf: its points are G)
    FnDomainMark<CV> g{n, xy, nullptr, out_xy, out_inf};
    bk.launch(g, n);
  }
  return rc;
}


template <class BK>
template <class CV>
int Engine<BK>::mul_fixed_chunk(size_t n, const u8* k, u8* out_xy, u8* out_inf) {
  typedef Work<CV> W;
  // a handful of items: the comb and the item's own inversion on a wave, one launch (the scalar is
  // Point#mul's: BYTES bytes as they stand -- coop_sign_point<CW, false>, no _truncateToN)
  constexpr bool row_k256 = CV::ENDO && W::L <= 8 && CoopK256::AVAILABLE;
  constexpr bool row_nist = !CV::ENDO && CoopNist<CV>::AVAILABLE;
  if constexpr ((row_k256 || row_nist) && W::NBYTES == W::BYTES) {
    if (n <= coop_grid_of<CV>() && out_inf) {
      typedef typename std::conditional<row_k256, CoopK256, CoopNist<CV>>::type CW;
      FnMulFixedC<CV, CW> fc{n, k, (const typename W::A*)comb_[CV::ID], out_xy, out_inf};
      bk.launch_coop(fc, n);
      return E_OK;
    }
  }
  u32* jac = (u32*)scratch(S_JAC, n * 3 * W::NS * 4);
  if (!jac) return fail(E_NOMEM, "scratch allocation failed");
  if (W::L > 12 && n > ELL_P521_PAIR_MIN) {
    FnMulFixed<CV, (W::L > 12 ? 2 : 0)> f{n, k, (const typename W::A*)comb_[CV::ID], jac};
    bk.launch(f, n);
  } else {
    FnMulFixed<CV> f{n, k, (const typename W::A*)comb_[CV::ID], jac};
    bk.launch(f, n);
  }
  return normalize_chunk<CV>(n, jac, out_xy, out_inf, nullptr);
}


template <class BK>
template <class CV>
int Engine<BK>::mul_add2_chunk(size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                   u8* out_xy, u8* out_inf) {
  typedef Work<CV> W;
  u32* jac = (u32*)scratch(S_JAC, n * 3 * W::NS * 4);
  typename W::J* tbl = (typename W::J*)scratch(S_TBL, n * 2 * W::TBLJ * sizeof(typename W::J));
  if (!tbl || !jac) return fail(E_NOMEM, "scratch allocation failed");
  FnMulAdd2<CV> f{n, k1, xy1, k2, xy2, tbl, jac};
  bk.launch(f, n);
  int rc = normalize_chunk<CV>(n, jac, out_xy, out_inf, nullptr);
  if (rc == E_OK && out_inf) {
    FnDomainMark<CV> g{n, xy1, xy2, out_xy, out_inf};
    bk.launch(g, n);
  }
  return rc;
}

template <class BK>
template <class CV>
int Engine<BK>::mul_add_g_chunk(size_t n, const u8* k1, const u8* k2, const u8* xy2, u8* out_xy,
                    u8* out_inf) {
  typedef Work<CV> W;
  if constexpr (CV::ENDO && W::L <= 8) {
    // (both gates, as in mul_var_chunk / ecdsa_chunk: ELLGPU_SMALL_GRID=0 keeps the call on the
    // full-grid tuning whatever the parted threshold says)
    if (n <= small_grid() && n <= parted_grid()) {   // most SIMDs would idle: k2's halves and k1's comb in three waves
      u32* pj = (u32*)scratch(S_JAC, 3 * n * 3 * W::NS * 4);
      typename W::VT* pt = (typename W::VT*)scratch(S_TBL, 2 * n * (size_t)W::template stride<true>() * sizeof(typename W::VT));
      if (!pt || !pj) return fail(E_NOMEM, "scratch allocation failed");
      if (!rows_for(n) && n <= coop_grid()) {
        FnMulPartsC fc{n, k2, xy2, pj, k1, (const typename W::A*)comb_[CV::ID]};
        bk.launch_coop(fc, 3 * n);
      } else if (rows_for(n)) {
        FnMulPartsR fr{n, k2, xy2, pj, k1, (const typename W::A*)comb_[CV::ID]};
        bk.launch_coop(fr, 3 * ((n + 3) / 4));
      } else {
      const size_t npad = (n + 127) & ~(size_t)127;
      FnMulParts<CV> fp{n, npad, k2, xy2, pt, pj, k1, (const typename W::A*)comb_[CV::ID]};
      launch_fn(fp, 2 * npad + n);
      }
      FnMulJoin<CV> fj{n, pj, true, xy2, out_xy, out_inf, nullptr};
      return launch_fn(fj, n);
    }
  }
  if constexpr (!CV::ENDO && CoopNist<CV>::AVAILABLE) {
    if (n <= coop_grid_of<CV>()) {                 // the ladder of k2 and the comb of k1 on a wave each, joined on one lane
      u32* pj = (u32*)scratch(S_JAC, 2 * n * 3 * W::NS * 4);
      if (!pj) return fail(E_NOMEM, "scratch allocation failed");
      FnMulPartsN<CV> fc{n, k2, xy2, pj, k1, (const typename W::A*)comb_[CV::ID]};
      bk.launch_coop(fc, 2 * n);
      FnMulJoin<CV> fj{n, pj, false, xy2, out_xy, out_inf, nullptr};
      return launch_fn(fj, n);
    }
  }
  u32* jac = (u32*)scratch(S_JAC, n * 3 * W::NS * 4);
  typename W::VT* tbl = (typename W::VT*)scratch(S_TBL, n * (size_t)W::template stride<false>() * sizeof(typename W::VT));
  if (!tbl || !jac) return fail(E_NOMEM, "scratch allocation failed");
  if (W::L > 12 && n > ELL_P521_PAIR_MIN) {
    FnMulAddG<CV, (W::L > 12 ? 2 : 0)> f{n, k1, k2, xy2, (const typename W::A*)comb_[CV::ID], tbl, jac};
    bk.launch(f, n);
  } else {
    FnMulAddG<CV> f{n, k1, k2, xy2, (const typename W::A*)comb_[CV::ID], tbl, jac};
    bk.launch(f, n);
  }
  int rc = normalize_chunk<CV>(n, jac, out_xy, out_inf, nullptr);
  if (rc == E_OK && out_inf) {
    FnDomainMark<CV> g{n, nullptr, xy2, out_xy, out_inf};
    bk.launch(g, n);
  }
  return rc;
}


template <class BK>
template <int U>
int Engine<BK>::edc_chunk(int op, size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                          const u8* a, const u8* b, u8* out_xy, u8* out_inf) {
  u32* proj = (u32*)scratch(S_JAC, n * 3 * 8 * 4);
  u32* pre = (u32*)scratch(S_PRE, n * 8 * 4);
  EdcWork::P* tbl = (EdcWork::P*)scratch(S_TBL, n * 16 * sizeof(EdcWork::P));
  if (!proj || !pre || !tbl) return fail(E_NOMEM, "scratch allocation failed");
  if (op == 0) {
    FnEdcMulVar f{n, k1, xy1, tbl, proj};
    bk.launch(f, n);
  } else if (op == 1) {
    FnEdcMulAdd2 f{n, k1, xy1, k2, xy2, tbl, proj};
    bk.launch(f, n);
  } else {
    FnEdcPointAdd f{n, xy1, a, xy2, b, proj};
    bk.launch(f, n);
  }
  const int K = norm_batch_for(n);
  size_t T = (n + K - 1) / K;
  FnEdcNormalize g{T, n, K, proj, pre, out_xy, out_inf};
  bk.launch(g, T);
  if (op != 2 && out_inf) {                       // Point#add is one formula: the reference's own
    FnEdcDomainMark h{n, xy1, op == 1 ? xy2 : nullptr, out_xy, out_inf};
    bk.launch(h, n);
  }
  return E_OK;
}

template <class BK>
template <class Fn>
int Engine<BK>::launch_fn(const Fn& f, size_t nthreads) {
  bk.launch(f, nthreads);
  return E_OK;
}

template <class BK>
template <class CV>
int Engine<BK>::ecdsa_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* r, const u8* s,
                const u8* pub, u8* ok, u8* st) {
  typedef Work<CV> W;
  bool wide = false;
  if constexpr (CV::ENDO && W::L <= 8) wide = n <= small_grid();
  const size_t slots = wide ? (size_t)W::template stride<true>() : (size_t)W::template stride<false>();
  typename W::VT* tbl = (typename W::VT*)scratch(S_TBL, n * slots * sizeof(typename W::VT));
  u32* pre = (u32*)scratch(S_PRE, n * (W::LN > W::NS ? W::LN : W::NS) * 4);
  u32* u12 = (u32*)scratch(S_U12, n * 2 * W::LN * 4);
  u8* valid = (u8*)scratch(S_VALID, n);
  if (!tbl || !pre || !u12 || !valid) return fail(E_NOMEM, "scratch allocation failed");
  const int K = inv_batch_for(n, INV_BATCH_N);
  size_t T = (n + K - 1) / K;
  FnEcdsaPrep<CV> f1{T, n, K, hash, hash_len, shift, r, s, pre, u12, valid};
  if constexpr (CV::ENDO && W::L <= 8 && ELL_SPLIT_SMALL_VERIFY) {
    // (the two-kernel form on a FULL grid was measured too, round 4: 8.22 -> 8.37 ms per 2^20 --
    // the table kernel alone takes 0.65 ms where building the table inside ecdsa_main costs
    // 0.5 ms, and there is no latency to hide at four waves per SIMD; small grids only)
    if (wide && split_small_verify()) {
      // latency-bound batch: the window table does not depend on s^-1, so it is built BESIDE
      // ecdsa_prep -- one launch whose first workgroups run the prep and whose others build the
      // tables (FnEcdsaPrepTable); the ladder follows in stream order.  (Round 4 first ran the two
      // as separate kernels on two streams with an event fork / join: 0.6-1.9 % slower than this
      // fused launch, profiles/r04_split_verify_ab.txt.)
      const int Ks = inv_batch_beside(n, INV_BATCH_N);
      const size_t Ts = (n + Ks - 1) / Ks;
      const size_t tpad = (Ts + 127) & ~(size_t)127;          // whole workgroups of either kind
      const bool rows = n <= parted_grid() && rows_for(n);                           // one item per row
      if (!rows && n <= parted_grid() && n <= coop_grid()) {
        // a handful of items: the prep on a wave per item, the table on another (row layer), one launch
        FnEcdsaPrepTableC fptc{n, hash, hash_len, shift, r, s, pre, u12, valid, pub, tbl};
        bk.launch_coop(fptc, 2 * n);
      } else if (rows) {
        const size_t pu = (n + 63) / 64;
        FnEcdsaPrepTableR fptr{n, pu, hash, hash_len, shift, r, s, pre, u12, valid, pub, tbl};
        bk.launch_coop(fptr, pu + (n + 3) / 4);
      } else {
        FnEcdsaPrepTable<CV> fpt{{Ts, n, Ks, hash, hash_len, shift, r, s, pre, u12, valid}, tpad, {n, pub, tbl}};
        launch_fn(fpt, tpad + n);
      }
      if (n <= parted_grid()) {
        // most SIMDs would idle beside this batch: three lanes per item (FnEcdsaParts), then the join
        u32* jac = (u32*)scratch(S_JAC, n * 3 * 3 * W::NS * 4);
        if (!jac) return fail(E_NOMEM, "scratch allocation failed");
        if (!rows && n <= coop_grid()) {
          // a handful of items: every part on a wave of its own, lanes-per-item arithmetic
          FnEcdsaPartsC fc{n, u12, (const typename W::A*)comb_[CV::ID], tbl, jac};
          bk.launch_coop(fc, 3 * n);
        } else if (rows) {
          FnEcdsaPartsR fr{n, u12, (const typename W::A*)comb_[CV::ID], tbl, jac};
          bk.launch_coop(fr, 3 * ((n + 3) / 4));
        } else {
        const size_t npad = (n + 127) & ~(size_t)127;          // whole workgroups per part
        FnEcdsaParts<CV> fp{n, npad, u12, (const typename W::A*)comb_[CV::ID], tbl, jac};
        launch_fn(fp, 2 * npad + n);
        }
        FnEcdsaJoin<CV> fj{n, valid, r, pub, tbl, jac, ok, st};
        return launch_fn(fj, n);
      }
      FnEcdsaLadder<CV, true> fl{n, u12, valid, r, pub, (const typename W::A*)comb_[CV::ID], tbl, ok, st};
      return launch_fn(fl, n);
    }
  }
  if constexpr (!CV::ENDO && CoopNist<CV>::AVAILABLE) {
    if (n <= coop_grid_of<CV>()) {
      // a handful of items on a curve without an endomorphism: the scalar-field prep as it is
      // (one inversion per item), then the ladder and the comb of every item on a WAVE each (the
      // row layer, coop_mont.h), and the one-lane join
      typedef CoopNist<CV> CW;
      u32* jac = (u32*)scratch(S_JAC, n * 2 * 3 * W::NS * 4);
      typename CW::A* gt = (typename CW::A*)scratch(S_TBL, n * CW::TABLE_BYTES);
      if (!jac || !gt) return fail(E_NOMEM, "scratch allocation failed");
      // the prep on a wave per item, the window table of the key on another, one launch ...
      FnEcdsaPrepTableN<CV> fpt{n, hash, hash_len, shift, r, s, pre, u12, valid, pub, gt};
      bk.launch_coop(fpt, 2 * n);
      // ... then the ladder over that table and the comb
      FnEcdsaPartsN<CV> fc{n, u12, pub, (const typename W::A*)comb_[CV::ID], jac, gt};
      bk.launch_coop(fc, 2 * n);
      FnEcdsaJoin2<CV> fj{n, valid, r, pub, jac, ok, st};
      return launch_fn(fj, n);
    }
  }
  launch_fn(f1, T);
  if (W::L > 12 && n > ELL_P521_PAIR_MIN) {
    FnEcdsaMain<CV, (W::L > 12 ? 2 : 0)> f2{n, u12, valid, r, pub, (const typename W::A*)comb_[CV::ID], tbl, ok, st};
    bk.launch(f2, n);
    return E_OK;
  }
  if constexpr (CV::ENDO && W::L <= 8) {
    if (wide) {                             // at most three waves per SIMD: the register-rich tuning
      FnEcdsaMain<CV, 3, true> f2{n, u12, valid, r, pub, (const typename W::A*)comb_[CV::ID], tbl, ok, st};
      return launch_fn(f2, n);
    }
  }
  FnEcdsaMain<CV> f2{n, u12, valid, r, pub, (const typename W::A*)comb_[CV::ID], tbl, ok, st};
  bk.launch(f2, n);
  return E_OK;
}

// ---- Edwards / Montgomery -------------------------------------------------

template <class BK>
template <int U>
int Engine<BK>::ensure_ed_comb() {
  if (comb_[CURVE_ED25519]) return E_OK;
  const size_t n = EdWork::COMB_ENTRIES;
  std::vector<u8> ks(n * 32, 0), pts(n * 64, 0);
  u8 g[64];
  {
    u32 gx[8], gy[8];
    for (int i = 0; i < 8; i++) { gx[i] = consts::ED25519_C::gx_plain[i]; gy[i] = consts::ED25519_C::gy_plain[i]; }
    store_be<8>(g, gx, 32);
    store_be<8>(g + 32, gy, 32);
  }
  for (int w = 0; w < EdWork::COMB_W; w++)
    for (int d = 1; d <= EdWork::COMB_DIG; d++) {
      size_t i = (size_t)w * EdWork::COMB_DIG + (d - 1);
      int byte0 = w * (EdWork::COMB_BITS / 8);
      for (int bb = 0; bb < EdWork::COMB_BITS / 8; bb++) ks[i * 32 + (31 - (byte0 + bb))] = (u8)(d >> (8 * bb));
      memcpy(&pts[i * 64], g, 64);
    }
  void* comb = bk.alloc(n * sizeof(EdWork::P));
  u8* dk = (u8*)bk.alloc(ks.size());
  u8* dp = (u8*)bk.alloc(pts.size());
  if (!comb || !dk || !dp) return fail(E_NOMEM, "comb table allocation failed");
  bk.h2d(dk, ks.data(), ks.size());
  bk.h2d(dp, pts.data(), pts.size());
  int rc = ed_mul_var_chunk(n, dk, dp, nullptr, nullptr, (EdWork::P*)comb);
  bk.sync();
  bk.free_(dk);
  bk.free_(dp);
  if (rc) { bk.free_(comb); return rc; }
  comb_[CURVE_ED25519] = comb;
  return E_OK;
}

template <class BK>
template <int U>
int Engine<BK>::ed_normalize_chunk(size_t n, const u32* ext, u8* out_xy, u8* out_inf, EdWork::P* raw) {
  u32* pre = (u32*)scratch(S_PRE, n * 8 * 4);
  if (!pre) return fail(E_NOMEM, "scratch allocation failed");
  const int K = norm_batch_for(n);
  size_t T = (n + K - 1) / K;
  FnEdNormalize f{T, n, K, ext, pre, out_xy, out_inf, raw};
  bk.launch(f, T);
  return E_OK;
}

template <class BK>
template <int U>
int Engine<BK>::ed_mul_var_chunk(size_t n, const u8* k, const u8* xy, u8* out_xy, u8* out_inf, EdWork::P* raw) {
  EdWork::P* tbl = (EdWork::P*)scratch(S_TBL, n * 8 * sizeof(EdWork::P));
  u32* ext = (u32*)scratch(S_JAC, n * 4 * 8 * 4);
  if (!tbl || !ext) return fail(E_NOMEM, "scratch allocation failed");
  if (n <= coop_grid()) {
    FnEdMulC fc{n, nullptr, nullptr, k, xy, nullptr, ext};
    bk.launch_coop(fc, n);
  } else {
    FnEdMulVar f{n, k, xy, tbl, ext};
    bk.launch(f, n);
  }
  int rc = ed_normalize_chunk(n, ext, out_xy, out_inf, raw);
  if (rc == E_OK && out_inf) {
    FnEdDomainMark g{n, xy, nullptr, out_xy, out_inf};
    bk.launch(g, n);
  }
  return rc;
}

template <class BK>
template <int U>
int Engine<BK>::ed_mul_fixed_chunk(size_t n, const u8* k, u8* out_xy, u8* out_inf) {
  if (n <= coop_grid() && out_xy) {                    // a handful of items: one item per wave, one launch
    FnEdMulFixedC fc{n, k, (const EdWork::P*)comb_[CURVE_ED25519], out_xy, out_inf};
    bk.launch_coop(fc, n);
    return E_OK;
  }
  u32* ext = (u32*)scratch(S_JAC, n * 4 * 8 * 4);
  if (!ext) return fail(E_NOMEM, "scratch allocation failed");
  FnEdMulFixed f{n, k, (const EdWork::P*)comb_[CURVE_ED25519], ext};
  bk.launch(f, n);
  return ed_normalize_chunk(n, ext, out_xy, out_inf, nullptr);
}

template <class BK>
template <int U>
int Engine<BK>::ed_mul_add2_chunk(size_t n, const u8* k1, const u8* xy1, const u8* k2, const u8* xy2,
                      u8* out_xy, u8* out_inf) {
  u32* ext = (u32*)scratch(S_JAC, n * 4 * 8 * 4);
  EdWork::P* tbl = (EdWork::P*)scratch(S_TBL, n * 16 * sizeof(EdWork::P));
  if (!tbl || !ext) return fail(E_NOMEM, "scratch allocation failed");
  if (n <= coop_grid()) {
    FnEdMulC fc{n, k1, xy1, k2, xy2, (const EdWork::P*)comb_[CURVE_ED25519], ext};
    bk.launch_coop(fc, n);
  } else if (xy1) {
    FnEdMulAdd2 f{n, k1, xy1, k2, xy2, tbl, ext};
    bk.launch(f, n);
  } else {
    FnEdMulAddG f{n, k1, k2, xy2, (const EdWork::P*)comb_[CURVE_ED25519], tbl, ext};
    bk.launch(f, n);
  }
  int rc = ed_normalize_chunk(n, ext, out_xy, out_inf, nullptr);
  if (rc == E_OK && out_inf) {
    FnEdDomainMark g{n, xy1, xy2, out_xy, out_inf};
    bk.launch(g, n);
  }
  return rc;
}

template <class BK>
template <int U>
int Engine<BK>::x25519_chunk(size_t n, const u8* k, const u8* x, u8* out_x, u8* out_inf, u8* out_bad) {
  u32* xz = (u32*)scratch(S_JAC, n * 2 * 8 * 4);
  u32* pre = (u32*)scratch(S_PRE, n * 8 * 4);
  if (!xz || !pre) return fail(E_NOMEM, "scratch allocation failed");
  if (n <= coop_grid()) {
    FnX25519C fc{n, k, x, xz, out_bad};
    bk.launch_coop(fc, out_bad ? 2 * n : n);
  } else {
    FnX25519 f{n, k, x, xz};
    bk.launch(f, n);
    if (out_bad) {
      FnX25519Validate fv{n, x, out_bad};
      bk.launch(fv, n);
    }
  }
  const int K = norm_batch_for(n);
  size_t T = (n + K - 1) / K;
  FnX25519Normalize g{T, n, K, xz, pre, out_x, out_inf};
  bk.launch(g, T);
  return E_OK;
}


template <class BK>
template <class CV>
int Engine<BK>::decompress_chunk(size_t n, const u8* x, const u8* odd, u8* out_xy, u8* out_ok) {
  if constexpr (!CV::F::HAS_SQRT) {
    return fail(E_UNSUPPORTED, "point decompression is not available in this field");
  } else {
    if constexpr (CV::ID == CURVE_SECP256K1 && CoopK256::AVAILABLE) {
      if (n <= coop_grid()) {                          // a handful of items: the square root on a wave each
        FnDecompressC fc{n, x, odd, out_xy, out_ok};
        bk.launch_coop(fc, n);
        return E_OK;
      }
    }
    FnDecompress<CV> f{n, x, odd, out_xy, out_ok};
    bk.launch(f, n);
    return E_OK;
  }
}
template <class BK>
template <class CV>
int Engine<BK>::codec_chunk(int op, size_t n, const u8* in, size_t len, int flag, const u8* inf, u8* out,
                            u8* status) {
  typedef Work<CV> W;
  if (op == OP_DECODE) {
    FnDecodePoint<CV> f{n, in, len, out, status};
    bk.launch(f, n);
    return E_OK;
  }
  if (op == OP_ENCODE) {
    FnEncodePoint<CV> f{n, in, flag, out};
    bk.launch(f, n);
    return E_OK;
  }
  // validate: curve equation, then (flag) n * P == O through the variable-base ladder
  u8* tmp = nullptr;
  if (flag) {
    tmp = (u8*)scratch(S_U12, n * (3 * (size_t)W::BYTES + 1));
    if (!tmp) return fail(E_NOMEM, "scratch allocation failed");
  }
  FnValidatePoint<CV> f{n, in, inf, status, tmp};
  bk.launch(f, n);
  if (!flag) return E_OK;
  u8* mxy = tmp + n * W::BYTES;
  u8* minf = mxy + n * 2 * W::BYTES;
  int rc = mul_var_chunk<CV>(n, tmp, in, mxy, minf, nullptr);
  if (rc) return rc;
  FnOrderStatus g{n, minf, status};
  bk.launch(g, n);
  return E_OK;
}
template <class BK>
template <class CV>
int Engine<BK>::point_add_chunk(size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2,
                                u8* out_xy, u8* out_inf) {
  typedef Work<CV> W;
  u32* jac = (u32*)scratch(S_JAC, n * 3 * W::NS * 4);
  if (!jac) return fail(E_NOMEM, "scratch allocation failed");
  FnPointAdd<CV> f{n, xy1, inf1, xy2, inf2, jac};
  bk.launch(f, n);
  return normalize_chunk<CV>(n, jac, out_xy, out_inf, nullptr);
}
template <class BK>
template <class CV>
int Engine<BK>::der_chunk(int op, size_t n, const u8* a, const u8* b, size_t stride, u32* lens, u8* o1, u8* o2,
                          u8* o3) {
  if (op == OP_FROM_DER) {
    FnSigFromDer<CV> f{n, a, stride, lens, o1, o2, o3};
    bk.launch(f, n);
  } else if (op == OP_TO_DER) {
    FnSigToDer<CV> f{n, a, b, o1, stride, lens};
    bk.launch(f, n);
  } else {
    FnWireStatus<CV> f{n, a, b, o3, o1, o2};
    bk.launch(f, n);
  }
  return E_OK;
}
template <class BK>
template <class CV>
int Engine<BK>::sign_det_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* priv,
                               int canonical, u8* out_r, u8* out_s, u8* out_recid, u8* out_ok) {
  typedef Work<CV> W;
  u8* nonces = (u8*)scratch(S_TBL, n * W::NBYTES);        // the sign pipeline uses S_JAC / S_U12 / S_PRE
  if (!nonces) return fail(E_NOMEM, "scratch allocation failed");
  FnDetNonce<CV> f{n, hash, hash_len, shift, priv, nonces};
  bk.launch(f, n);
  return sign_chunk<CV>(n, hash, hash_len, shift, priv, nonces, canonical, out_r, out_s, out_recid, out_ok);
}

template <class BK>
template <class CV>
int Engine<BK>::recover_chunk(size_t n, const u8* hash, int hash_len, const u8* r, const u8* s,
                              const u8* recid, u8* out_xy, u8* out_status) {
  if constexpr (!CV::F::HAS_SQRT) {
    return fail(E_UNSUPPORTED, "public-key recovery needs point decompression, which this field does not have");
  } else {
    typedef Work<CV> W;
    const size_t B = W::BYTES, NB = W::NBYTES;
    // scalars, the x-coordinates and the decompressed points R live across the ladder kernels,
    // which use S_TBL / S_JAC / S_PRE only
    u8* buf = (u8*)scratch(S_U12, n * (2 * NB + 3 * B));
    u8* flags = (u8*)scratch(S_VALID, 3 * n);
    u32* pre = (u32*)scratch(S_PRE, n * W::LN * 4);
    if (!buf || !flags || !pre) return fail(E_NOMEM, "scratch allocation failed");
    u8* s1 = buf;
    u8* s2 = s1 + n * NB;
    u8* xs = s2 + n * NB;
    u8* rxy = xs + n * B;
    u8* odd = flags;
    u8* dec_ok = flags + n;
    u8* inf = flags + 2 * n;
    int rc = E_OK;
    bool front = false;
    if constexpr (CV::ID == CURVE_SECP256K1 && CoopK256::AVAILABLE) {
      if (n <= coop_grid()) {                          // a handful of items: R's square root beside r^-1, one launch
        FnRecoverPartsC fc{n, hash, hash_len, r, s, recid, pre, xs, odd, s1, s2, out_status, rxy, dec_ok};
        bk.launch_coop(fc, 2 * n);
        front = true;
      }
    }
    if (!front) {
      const int Kr = inv_batch_for(n, INV_BATCH_N);
      size_t T = (n + Kr - 1) / Kr;
      FnRecoverPrep<CV> f1{T, n, Kr, hash, hash_len, r, s, recid, pre, xs, odd, s1, s2, out_status};
      launch_fn(f1, T);
      rc = decompress_chunk<CV>(n, xs, odd, rxy, dec_ok);
      if (rc) return rc;
    }
    rc = mul_add_g_chunk<CV>(n, s1, s2, rxy, out_xy, inf);         // s1 * G + s2 * R
    if (rc) return rc;
    FnRecoverFinish<CV> f2{n, dec_ok, inf, out_xy, out_status};
    bk.launch(f2, n);
    return E_OK;
  }
}

template <class BK>
template <int U>
int Engine<BK>::ed_decompress_chunk(size_t n, const u8* y, const u8* odd, u8* out_xy, u8* out_ok) {
  FnEdDecompress f{n, y, odd, out_xy, out_ok};
  bk.launch(f, n);
  return E_OK;
}

template <class BK>
template <int U>
int Engine<BK>::ed_point_add_chunk(size_t n, const u8* xy1, const u8* inf1, const u8* xy2, const u8* inf2,
                                   u8* out_xy, u8* out_inf) {
  u32* ext = (u32*)scratch(S_JAC, n * 4 * 8 * 4);
  if (!ext) return fail(E_NOMEM, "scratch allocation failed");
  FnEdPointAdd f{n, xy1, inf1, xy2, inf2, ext};
  bk.launch(f, n);
  return ed_normalize_chunk(n, ext, out_xy, out_inf, nullptr);
}

template <class BK>
template <int U>
int Engine<BK>::ed_codec_chunk(int op, size_t n, const u8* in, int flag, const u8* inf, u8* out, u8* status) {
  if (op == OP_DECODE) {
    FnEdDecodePoint f{n, in, out, status};
    bk.launch(f, n);
    return E_OK;
  }
  if (op == OP_ENCODE) {
    FnEdEncodePoint f{n, in, out};
    bk.launch(f, n);
    return E_OK;
  }
  u8* tmp = nullptr;
  if (flag) {
    tmp = (u8*)scratch(S_U12, n * (3 * 32 + 1));
    if (!tmp) return fail(E_NOMEM, "scratch allocation failed");
  }
  FnEdValidatePoint f{n, in, inf, status, tmp};
  bk.launch(f, n);
  if (!flag) return E_OK;
  u8* mxy = tmp + n * 32;
  u8* minf = mxy + n * 64;
  int rc = ed_mul_var_chunk(n, tmp, in, mxy, minf, nullptr);
  if (rc) return rc;
  FnOrderStatus g{n, minf, status};
  bk.launch(g, n);
  return E_OK;
}

template <class BK>
template <int U>
int Engine<BK>::eddsa_chunk(size_t n, size_t o, const u8* msgs, const u64* off, size_t msg_len,
                            const u8* sigs, const u8* pubs, u8* ok, u8* err) {
  if (n <= coop_grid()) {
    u32* ext = (u32*)scratch(S_JAC, 2 * n * 3 * 8 * 4);
    u8* flags = (u8*)scratch(S_VALID, 2 * n);
    if (!ext || !flags) return fail(E_NOMEM, "scratch allocation failed");
    FnEddsaPartsC fc{n, off ? msgs : msgs + o * msg_len, off ? off + o : nullptr, msg_len, sigs + o * 64,
                     pubs + o * 32, (const EdWork::P*)comb_[CURVE_ED25519], ext, flags};
    bk.launch_coop(fc, 2 * n);
    FnEddsaJoin fj{n, ext, flags, ok + o, err ? err + o : nullptr};
    bk.launch(fj, n);
    return E_OK;
  }
  EdWork::P* tbl = (EdWork::P*)scratch(S_TBL, n * 8 * sizeof(EdWork::P));
  if (!tbl) return fail(E_NOMEM, "scratch allocation failed");
  FnEddsaVerify f{n, off ? msgs : msgs + o * msg_len, off ? off + o : nullptr, msg_len,
                  sigs + o * 64, pubs + o * 32, (const EdWork::P*)comb_[CURVE_ED25519], tbl,
                  ok + o, err ? err + o : nullptr};
  bk.launch(f, n);
  return E_OK;
}

template <class BK>
template <int U>
int Engine<BK>::eddsa_sign_chunk(size_t n, size_t o, const u8* secrets, const u8* msgs, const u64* off,
                                 size_t msg_len, u8* sig, u8* pub) {
  u8* scal = (u8*)scratch(S_U12, 2 * n * 32);                     // a and r, big-endian
  u8* xy = (u8*)scratch(S_TBL, 2 * n * 64);                       // A = a G and R = r G, affine
  u8* inf = (u8*)scratch(S_VALID, 2 * n);
  if (!scal || !xy || !inf) return fail(E_NOMEM, "scratch allocation failed");
  const u8* m0 = off ? msgs : msgs + o * msg_len;
  const u64* off0 = off ? off + o : nullptr;
  FnEddsaSignPre f1{n, secrets + o * 32, m0, off0, msg_len, scal};
  bk.launch(f1, n);
  int rc = ed_mul_fixed_chunk(2 * n, scal, xy, inf);
  if (rc) return rc;
  FnEddsaSignPost f2{n, m0, off0, msg_len, scal, xy, sig + o * 64, pub ? pub + o * 32 : nullptr};
  bk.launch(f2, n);
  return E_OK;
}

template <class BK>
template <class CV>
int Engine<BK>::sign_chunk(size_t n, const u8* hash, int hash_len, int shift, const u8* priv,
                           const u8* nonces, int canonical, u8* out_r, u8* out_s, u8* out_recid,
                           u8* out_ok) {
  typedef Work<CV> W;
  u32* jac = (u32*)scratch(S_JAC, n * 3 * W::NS * 4);
  u8* kg = (u8*)scratch(S_U12, n * (2 * W::BYTES + 1));          // k*G affine + infinity flags
  if (!jac || !kg) return fail(E_NOMEM, "scratch allocation failed");
  u8* kg_inf = kg + n * 2 * W::BYTES;
  // a handful of items: k*G (comb + its own inversion) and k^-1 mod n on a wave each, one launch;
  // then the finish without an inversion of its own
  constexpr bool row_k256 = CV::ENDO && W::L <= 8 && CoopK256::AVAILABLE;
  constexpr bool row_nist = !CV::ENDO && CoopNist<CV>::AVAILABLE;
  if constexpr (row_k256 || row_nist) {
    if (n <= coop_grid_of<CV>()) {
      u32* kinv = (u32*)scratch(S_PRE, n * 2 * (W::LN > W::NS ? W::LN : W::NS) * 4);
      if (!kinv) return fail(E_NOMEM, "scratch allocation failed");
      u32* pre2 = kinv + n * (W::LN > W::NS ? W::LN : W::NS);
      typedef typename std::conditional<row_k256, CoopK256, CoopNist<CV>>::type CW;
      FnSignPartsC<CV, CW> fc{n, nonces, (const typename W::A*)comb_[CV::ID], kg, kg_inf, kinv};
      bk.launch_coop(fc, 2 * n);
      FnSignFinish<CV> f2{n, n, 1, hash, hash_len, shift, priv, nonces, kg, kg_inf, canonical, pre2,
                          out_r, out_s, out_recid, out_ok, kinv};
      return launch_fn(f2, n);
    }
  }
  if (W::L > 12 && n > ELL_P521_PAIR_MIN) {
    FnSignMul<CV, (W::L > 12 ? 2 : 0)> f1{n, nonces, (const typename W::A*)comb_[CV::ID], jac};
    bk.launch(f1, n);
  } else {
    FnSignMul<CV> f1{n, nonces, (const typename W::A*)comb_[CV::ID], jac};
    bk.launch(f1, n);
  }
  int rc = normalize_chunk<CV>(n, jac, kg, kg_inf, nullptr);
  if (rc) return rc;
  u32* pre = (u32*)scratch(S_PRE, n * (W::LN > W::NS ? W::LN : W::NS) * 4);
  if (!pre) return fail(E_NOMEM, "scratch allocation failed");
  const int Kf = inv_batch_for(n, INV_BATCH_N);
  size_t T = (n + Kf - 1) / Kf;
  FnSignFinish<CV> f2{T, n, Kf, hash, hash_len, shift, priv, nonces, kg, kg_inf, canonical, pre,
                      out_r, out_s, out_recid, out_ok, nullptr};
  launch_fn(f2, T);
  return E_OK;
}

}  // namespace ell
