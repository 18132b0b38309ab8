// ellgpu -- one translation unit per (curve, operation group): explicit
// instantiation of the Engine<HipBackend> member that launches the kernels.
//   hipcc -c inst.hip -DELL_INST_CURVE=CvP384 -DELL_INST_GROUP=4   (see build.py)
#include "engine_extern.h"

namespace ell {
#define ELL_NOKW
#if ELL_INST_GROUP == 0
ELL_DECL_G0(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 1
ELL_DECL_G1(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 2
ELL_DECL_G2(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 3
ELL_DECL_G3(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 4
ELL_DECL_G4(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 5
ELL_DECL_G5(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 6
ELL_DECL_G6(ELL_NOKW, ELL_INST_CURVE)
#elif ELL_INST_GROUP == 7
ELL_DECL_G7(ELL_NOKW)
#elif ELL_INST_GROUP == 8
ELL_DECL_G8(ELL_NOKW)
#elif ELL_INST_GROUP == 13
ELL_DECL_ED2(ELL_NOKW)
#elif ELL_INST_GROUP == 14
ELL_DECL_ED3(ELL_NOKW)
#elif ELL_INST_GROUP == 15
ELL_DECL_ED4(ELL_NOKW)
#elif ELL_INST_GROUP == 10
ELL_DECL_ED0(ELL_NOKW)
#elif ELL_INST_GROUP == 11
ELL_DECL_ED1(ELL_NOKW)
#elif ELL_INST_GROUP == 12
ELL_DECL_X(ELL_NOKW)
#elif ELL_INST_GROUP == 16
// the custom-curve kernels and the parameter block they read live in this one code object
__constant__ RtField g_rt;
int rt_upload_device(const RtField* f) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_rt), f, sizeof(RtField), 0, hipMemcpyHostToDevice);
}
ELL_DECL_CUSTOM(ELL_NOKW)
#else
#error "unknown ELL_INST_GROUP"
#endif
}  // namespace ell
