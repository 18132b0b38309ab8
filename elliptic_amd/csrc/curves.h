// ellgpu -- curve descriptions (compile-time traits) for the reference's
// short-Weierstrass presets (lib/elliptic/curves.js:43-134,176-206).
#pragma once

#include "curve_consts.h"
#include "fp.h"

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

struct CvSecp256k1 {
  typedef FpK256 F;
  typedef FpMont<consts::SECP256K1_N> Fn;
  typedef consts::SECP256K1_C C;
  static constexpr int A_KIND = 0;
  static constexpr bool ENDO = true;
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
  static constexpr int ID = ID_;
};

typedef CvNist<FpSolinas<SolP192>, consts::P192_N, consts::P192_C, CURVE_P192> CvP192;
typedef CvNist<FpSolinas<SolP224>, consts::P224_N, consts::P224_C, CURVE_P224> CvP224;
typedef CvNist<FpSolinas<SolP256>, consts::P256_N, consts::P256_C, CURVE_P256> CvP256;
typedef CvNist<FpSolinas<SolP384>, consts::P384_N, consts::P384_C, CURVE_P384> CvP384;
typedef CvNist<FpP521, consts::P521_N, consts::P521_C, CURVE_P521> CvP521;

}  // namespace ell
