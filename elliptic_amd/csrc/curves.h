// ellgpu -- curve descriptions (compile-time traits) for the reference's
// short-Weierstrass presets (lib/elliptic/curves.js:43-134,176-206).
#pragma once

#include "curve_consts.h"
#include "fp.h"
#include "fpk256l.h"
#include "fp_rt.h"

namespace ell {

// C-ABI curve ids (include/ellgpu.h)
enum CurveId {
  CURVE_SECP256K1 = 0,
  CURVE_P192 = 1,
  CURVE_P224 = 2,
  CURVE_P256 = 3,
  CURVE_P384 = 4,
  CURVE_P521 = 5,
  CURVE_ED25519 = 6,
  CURVE_CURVE25519 = 7,
  CURVE_COUNT = 8
};

// ELL_K256_LAZY = 1: the secp256k1 base field is the 9 x 29-bit signed-limb field of fpk256l.h
// (carry-free column products, lazy additions, two-product multiplies); 0 (default): the
// saturated 8 x 32-bit field of fp.h.  Both are bit-exact on the whole GPU suite.  MEASURED,
// round 2, same boxes (profiles/r02_lazy_field_*): the lazy field's operations are faster in
// isolation (doubling 923 vs 1038 issue units, mixed addition 1441 vs 1792, 148 instead of 168
// VGPRs, no scratch) but ecdsa_main as a whole is 4-5 % SLOWER (8.83 vs 8.49 ms, 9.15 vs 8.68 ms):
// it needs only 1.7 % fewer GPU cycles (GRBM_GUI_ACTIVE 19.68 M vs 20.02 M per launch) and its
// multiply-heavier mix (164 k instead of 117 k v_mad per verify) runs at a 7 % lower sustained
// clock (2.15 vs 2.31 GHz).  Kept as a build switch; DESIGN.md section 9.
#ifndef ELL_K256_LAZY
#define ELL_K256_LAZY 0
#endif
struct CvSecp256k1 {
#if ELL_K256_LAZY
  typedef FpK256L F;
#else
  typedef FpK256 F;
#endif
  typedef FpMont<consts::SECP256K1_N> Fn;
  typedef consts::SECP256K1_C C;
  static constexpr int A_KIND = 0;
  static constexpr bool ENDO = true;
  static constexpr bool JTABLE = false;
  static constexpr int ID = CURVE_SECP256K1;
};

// NIST curves (a = -3): FIELD is the base-field arithmetic -- the generalised-Mersenne fold for
// p192 / p224 / p256 / p384, the Mersenne fold for p521; the order field is always Montgomery.
template <class FIELD, class CN, class CC, int ID_>
struct CvNist {
  typedef FIELD F;
  typedef FpMont<CN> Fn;
  typedef CC C;
  static constexpr int A_KIND = 3;
  static constexpr bool ENDO = false;
  static constexpr bool JTABLE = false;
  static constexpr int ID = ID_;
};

// User-defined short Weierstrass curve over a run-time prime (fp_rt.h): arbitrary a, no
// endomorphism, no fixed-base tables, Jacobian window tables (JTABLE: the effective-affine table
// of the presets would need an inversion per item to get back to a curve whose a the doubling
// knows).  Only the scalar-multiplication kernels are instantiated for it.
namespace consts {
struct CUSTOM_C {
  static constexpr int L = 8;
  static constexpr int LN = 8;
  static constexpr int BYTES = 32;      // scalars and coordinates are 32-byte big-endian whatever p's size
  static constexpr int NBYTES = 32;
  static constexpr int NBITS = 256;
  static constexpr int PBITS = 256;
  static constexpr int A_KIND = 1;
  static constexpr u32 n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr u32 p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr u32 gx_plain[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr u32 gy_plain[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  static constexpr u32 b_plain[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
}  // namespace consts
struct CvCustom {
  typedef FpMontRT F;
  typedef FpMont<consts::SECP256K1_N> Fn;        // unused (no ECDSA on custom curves)
  typedef consts::CUSTOM_C C;
  static constexpr int A_KIND = 1;
  static constexpr bool ENDO = false;
  static constexpr bool JTABLE = true;
  static constexpr int ID = 16;
};
constexpr int CURVE_CUSTOM0 = 16;                 // C-ABI ids 16..31: ellgpu_curve_define_short / _edwards
constexpr int CURVE_CUSTOM_MAX = 16;

typedef CvNist<FpSolinas<SolP192>, consts::P192_N, consts::P192_C, CURVE_P192> CvP192;
typedef CvNist<FpSolinas<SolP224>, consts::P224_N, consts::P224_C, CURVE_P224> CvP224;
typedef CvNist<FpSolinas<SolP256>, consts::P256_N, consts::P256_C, CURVE_P256> CvP256;
typedef CvNist<FpSolinas<SolP384>, consts::P384_N, consts::P384_C, CURVE_P384> CvP384;
typedef CvNist<FpP521, consts::P521_N, consts::P521_C, CURVE_P521> CvP521;

}  // namespace ell
