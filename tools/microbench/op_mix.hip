// One field / group operation per kernel, for STATIC instruction-mix counts (tools/static_mix.py
// compiles this with --cuda-device-only -S -DOPMIX_CURVE=<Cv...> and counts the instructions of
// each kernel's main path).  Not part of the library.
#include "engine_extern.h"
using namespace ell;
typedef OPMIX_CURVE CV;
typedef CV::F F;
typedef F::El El;
typedef ShortOps<CV> G;
extern "C" __global__ void k_mul(El* p) { size_t i = threadIdx.x; p[i] = F::mul(p[i], p[i + 64]); }
extern "C" __global__ void k_sqr(El* p) { size_t i = threadIdx.x; p[i] = F::sqr(p[i]); }
extern "C" __global__ void k_add(El* p) { size_t i = threadIdx.x; p[i] = F::add(p[i], p[i + 64]); }
extern "C" __global__ void k_sub(El* p) { size_t i = threadIdx.x; p[i] = F::sub(p[i], p[i + 64]); }
extern "C" __global__ void k_ldst(El* p) { size_t i = threadIdx.x; p[i] = p[i + 64]; }
