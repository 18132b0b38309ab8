// ellgpu -- explicit-instantiation plumbing so that the (curve x operation)
// kernels compile in separate translation units (inst.hip, built in parallel by
// elliptic_amd/build.py) instead of one 20-minute TU.
#pragma once

#include "hip_backend.h"

namespace ell {

#define ELL_FOR_SHORT_CURVES(X) X(CvSecp256k1) X(CvP192) X(CvP224) X(CvP256) X(CvP384) X(CvP521)

#define ELL_DECL_G0(KW, CV)                                                                        \
  KW template int Engine<HipBackend>::mul_var_chunk<CV>(size_t, const u8*, const u8*, u8*, u8*,    \
                                                        Work<CV>::A*);                             \
  KW template int Engine<HipBackend>::normalize_chunk<CV>(size_t, const u32*, u8*, u8*,            \
                                                          Work<CV>::A*);                           \
  KW template int Engine<HipBackend>::ensure_comb<CV>();
#define ELL_DECL_G1(KW, CV) \
  KW template int Engine<HipBackend>::mul_fixed_chunk<CV>(size_t, const u8*, u8*, u8*);
#define ELL_DECL_G2(KW, CV)                                                                  \
  KW template int Engine<HipBackend>::mul_add2_chunk<CV>(size_t, const u8*, const u8*,       \
                                                         const u8*, const u8*, u8*, u8*);
#define ELL_DECL_G3(KW, CV)                                                                   \
  KW template int Engine<HipBackend>::mul_add_g_chunk<CV>(size_t, const u8*, const u8*,       \
                                                          const u8*, u8*, u8*);
#define ELL_DECL_G4(KW, CV)                                                                       \
  KW template int Engine<HipBackend>::ecdsa_chunk<CV>(size_t, const u8*, int, int, const u8*,     \
                                                      const u8*, const u8*, u8*, u8*);

#define ELL_DECL_G5(KW, CV)                                                                       \
  KW template int Engine<HipBackend>::decompress_chunk<CV>(size_t, const u8*, const u8*, u8*, u8*); \
  KW template int Engine<HipBackend>::codec_chunk<CV>(int, size_t, const u8*, size_t, int,         \
                                                      const u8*, u8*, u8*);                         \
  KW template int Engine<HipBackend>::point_add_chunk<CV>(size_t, const u8*, const u8*, const u8*, \
                                                          const u8*, u8*, u8*);                     \
  KW template int Engine<HipBackend>::der_chunk<CV>(int, size_t, const u8*, const u8*, size_t,     \
                                                    u32*, u8*, u8*, u8*);                           \
  KW template int Engine<HipBackend>::sign_chunk<CV>(size_t, const u8*, int, int, const u8*,       \
                                                     const u8*, int, u8*, u8*, u8*, u8*);           \
  KW template int Engine<HipBackend>::recover_chunk<CV>(size_t, const u8*, int, const u8*,         \
                                                        const u8*, const u8*, u8*, u8*);           \
  KW template int Engine<HipBackend>::sign_det_chunk<CV>(size_t, const u8*, int, int, const u8*,   \
                                                         int, u8*, u8*, u8*, u8*);
// scalar-field kernels (batched inversion / Montgomery arithmetic mod n): their own translation
// units, compiled with -mllvm -amdgpu-sched-strategy=max-ilp (build.py)
#define ELL_DECL_G6(KW, CV)                                                                        \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaPrep<CV>>(const FnEcdsaPrep<CV>&, size_t);   \
  KW template int Engine<HipBackend>::launch_fn<FnSignFinish<CV>>(const FnSignFinish<CV>&, size_t); \
  KW template int Engine<HipBackend>::launch_fn<FnRecoverPrep<CV>>(const FnRecoverPrep<CV>&, size_t);
// the small-grid (WIDE) verify kernel of the endomorphism curve: its own translation unit
#define ELL_DECL_G7(KW)                                                                              \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaMain<CvSecp256k1, 3, true>>(                  \
      const FnEcdsaMain<CvSecp256k1, 3, true>&, size_t);                                             \
  KW template int Engine<HipBackend>::launch_fn<FnMulVar<CvSecp256k1, 3, true>>(                     \
      const FnMulVar<CvSecp256k1, 3, true>&, size_t);                                                \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaPrepTable<CvSecp256k1>>(                      \
      const FnEcdsaPrepTable<CvSecp256k1>&, size_t);                                                 \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaLadder<CvSecp256k1, true>>(                   \
      const FnEcdsaLadder<CvSecp256k1, true>&, size_t);
// the parted verify (three lanes per item: batches that leave most of the device idle)
#define ELL_DECL_G8(KW)                                                                              \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaParts<CvSecp256k1>>(                          \
      const FnEcdsaParts<CvSecp256k1>&, size_t);                                                     \
  KW template int Engine<HipBackend>::launch_fn<FnEcdsaJoin<CvSecp256k1>>(                           \
      const FnEcdsaJoin<CvSecp256k1>&, size_t);                                                      \
  KW template int Engine<HipBackend>::launch_fn<FnMulParts<CvSecp256k1>>(                            \
      const FnMulParts<CvSecp256k1>&, size_t);                                                       \
  KW template int Engine<HipBackend>::launch_fn<FnMulJoin<CvSecp256k1>>(                             \
      const FnMulJoin<CvSecp256k1>&, size_t);
// user-defined short curves (CvCustom): scalar multiplication and point addition only
#define ELL_DECL_CUSTOM(KW)                                                                          \
  KW template int Engine<HipBackend>::mul_var_chunk<CvCustom>(size_t, const u8*, const u8*, u8*, u8*, \
                                                              Work<CvCustom>::A*);                   \
  KW template int Engine<HipBackend>::normalize_chunk<CvCustom>(size_t, const u32*, u8*, u8*,        \
                                                                Work<CvCustom>::A*);                 \
  KW template int Engine<HipBackend>::mul_add2_chunk<CvCustom>(size_t, const u8*, const u8*,         \
                                                               const u8*, const u8*, u8*, u8*);      \
  KW template int Engine<HipBackend>::point_add_chunk<CvCustom>(size_t, const u8*, const u8*, const u8*, \
                                                                const u8*, u8*, u8*);                \
  KW template int Engine<HipBackend>::edc_chunk<0>(int, size_t, const u8*, const u8*, const u8*,     \
                                                   const u8*, const u8*, const u8*, u8*, u8*);
#define ELL_DECL_ED2(KW)                                                                            \
  KW template int Engine<HipBackend>::ed_decompress_chunk<0>(size_t, const u8*, const u8*, u8*, u8*); \
  KW template int Engine<HipBackend>::ed_codec_chunk<0>(int, size_t, const u8*, int, const u8*, u8*, u8*); \
  KW template int Engine<HipBackend>::ed_point_add_chunk<0>(size_t, const u8*, const u8*, const u8*,  \
                                                            const u8*, u8*, u8*);
#define ELL_DECL_ED3(KW)                                                                          \
  KW template int Engine<HipBackend>::eddsa_chunk<0>(size_t, size_t, const u8*, const u64*, size_t, \
                                                     const u8*, const u8*, u8*, u8*);
#define ELL_DECL_ED4(KW)                                                                          \
  KW template int Engine<HipBackend>::eddsa_sign_chunk<0>(size_t, size_t, const u8*, const u8*,   \
                                                          const u64*, size_t, u8*, u8*);

#define ELL_DECL_ED0(KW)                                                                          \
  KW template int Engine<HipBackend>::ensure_ed_comb<0>();                                        \
  KW template int Engine<HipBackend>::ed_normalize_chunk<0>(size_t, const u32*, u8*, u8*,         \
                                                            EdWork::P*);                          \
  KW template int Engine<HipBackend>::ed_mul_var_chunk<0>(size_t, const u8*, const u8*, u8*, u8*, \
                                                          EdWork::P*);
#define ELL_DECL_ED1(KW)                                                                     \
  KW template int Engine<HipBackend>::ed_mul_fixed_chunk<0>(size_t, const u8*, u8*, u8*);    \
  KW template int Engine<HipBackend>::ed_mul_add2_chunk<0>(size_t, const u8*, const u8*,     \
                                                           const u8*, const u8*, u8*, u8*);
#define ELL_DECL_X(KW) \
  KW template int Engine<HipBackend>::x25519_chunk<0>(size_t, const u8*, const u8*, u8*, u8*, u8*);

// everything is extern by default ...
#define ELL_EXT_ALL(CV) \
  ELL_DECL_G0(extern, CV) ELL_DECL_G1(extern, CV) ELL_DECL_G2(extern, CV) ELL_DECL_G3(extern, CV) ELL_DECL_G4(extern, CV) \
  ELL_DECL_G5(extern, CV) ELL_DECL_G6(extern, CV)
ELL_FOR_SHORT_CURVES(ELL_EXT_ALL)
ELL_DECL_ED0(extern)
ELL_DECL_ED1(extern)
ELL_DECL_X(extern)
ELL_DECL_ED2(extern)
ELL_DECL_ED3(extern)
ELL_DECL_ED4(extern)
ELL_DECL_CUSTOM(extern)
ELL_DECL_G7(extern)
ELL_DECL_G8(extern)

}  // namespace ell
