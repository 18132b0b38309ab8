#!/usr/bin/env python3
"""Generates tools/microbench/issue_runs.hip: does the order of instruction CLASSES change what a
VALU instruction costs on gfx950?  profiles/r03_valu_patterns.log says plain VOP1/VOP2 ops
(v_mov_b32, v_add_u32) issue in ~2.5 cycles per wave64 in uninterrupted runs and ~4.2 when they
alternate one-to-one with v_mad_u64_u32.  This bench varies the RUN LENGTH: k multiplies (or
carry-class adds) followed by k plain ops, k = 1, 2, 4, 8, 16, and the 4:1 mixes the field
multiply really has, and prints the implied cost of the plain op in each arrangement.

    python tools/microbench/gen_issue_runs.py > tools/microbench/issue_runs.hip
    hipcc --offload-arch=gfx950 -O3 -o tools/microbench/_build/issue_runs tools/microbench/issue_runs.hip
"""
HEAVY = {
    "mad": lambda i: "v_mad_u64_u32 %%%d, s[20:21], %%16, %%17, %%%d" % (i % 8, i % 8),
    "addc": lambda i: "v_addc_co_u32_e64 %%%d, s[22:23], 0, %%%d, s[24:25]" % (8 + i % 8, 8 + i % 8),
    "cnd": lambda i: "v_cndmask_b32_e64 %%%d, %%%d, %%16, s[24:25]" % (8 + i % 8, 8 + i % 8),
}
PLAIN = {
    "mov": lambda i: "v_mov_b32 %%%d, %%%d" % (8 + i % 8, 8 + (i + 3) % 8),
    "add": lambda i: "v_add_u32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "xor": lambda i: "v_xor_b32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "lshl": lambda i: "v_lshlrev_b32 %%%d, 1, %%%d" % (8 + i % 8, 8 + i % 8),
}


def pattern(spec):
    """spec: list of (class name, count) -> (asm lines, n_heavy, n_plain)"""
    lines, nh, npl, ih, ip = [], 0, 0, 0, 0
    for name, cnt in spec:
        for _ in range(cnt):
            if name in HEAVY:
                lines.append(HEAVY[name](ih)); ih += 1; nh += 1
            else:
                lines.append(PLAIN[name](ip)); ip += 1; npl += 1
    return lines, nh, npl


# opcode classification: which opcodes are in the FAST class (~2.5 cycles per wave64 in a run of
# their own), and which of them ride for free behind a slow-class op (alternating with v_addc)?
CLASSIFY = {
    "v_sub_u32": lambda i: "v_sub_u32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_and_b32": lambda i: "v_and_b32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_or_b32": lambda i: "v_or_b32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_not_b32": lambda i: "v_not_b32 %%%d, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_lshrrev_b32": lambda i: "v_lshrrev_b32 %%%d, 3, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_min_u32": lambda i: "v_min_u32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_add3_u32": lambda i: "v_add3_u32 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_lshl_add_u32": lambda i: "v_lshl_add_u32 %%%d, %%16, 2, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_and_or_b32": lambda i: "v_and_or_b32 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_xad_u32": lambda i: "v_xad_u32 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_bfe_u32": lambda i: "v_bfe_u32 %%%d, %%%d, 3, 8" % (8 + i % 8, 8 + i % 8),
    "v_bfi_b32": lambda i: "v_bfi_b32 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_perm_b32": lambda i: "v_perm_b32 %%%d, %%16, %%%d, %%17" % (8 + i % 8, 8 + i % 8),
    "v_alignbit_b32": lambda i: "v_alignbit_b32 %%%d, %%16, %%%d, 5" % (8 + i % 8, 8 + i % 8),
    "v_mul_lo_u32": lambda i: "v_mul_lo_u32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_mul_hi_u32": lambda i: "v_mul_hi_u32 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_mul_u32_u24": lambda i: "v_mul_u32_u24 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_mad_u32_u24": lambda i: "v_mad_u32_u24 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_add_co_u32 e32 (vcc out)": lambda i: "v_add_co_u32_e32 %%%d, vcc, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_addc_co_u32 e32 (vcc in+out)": lambda i: "v_addc_co_u32_e32 %%%d, vcc, %%16, %%%d, vcc" % (8 + i % 8, 8 + i % 8),
    "v_add_co_u32 e64 (sgpr out)": lambda i: "v_add_co_u32_e64 %%%d, s[22:23], %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_cmp_lt_u32 e64": lambda i: "v_cmp_lt_u32_e64 s[22:23], %%16, %%%d" % (8 + i % 8),
    "v_cmp_lt_u32 e32": lambda i: "v_cmp_lt_u32_e32 vcc, %%16, %%%d" % (8 + i % 8),
    "v_cndmask_b32 e32 (vcc)": lambda i: "v_cndmask_b32_e32 %%%d, %%16, %%%d, vcc" % (8 + i % 8, 8 + i % 8),
    "v_pk_add_u16": lambda i: "v_pk_add_u16 %%%d, %%16, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_pk_mad_u16": lambda i: "v_pk_mad_u16 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_pk_mov_b32": lambda i: "v_pk_mov_b32 %%%d, %%%d, %%%d" % (i % 8, (i + 1) % 8, (i + 2) % 8),
    "v_lshl_add_u64": lambda i: "v_lshl_add_u64 %%%d, %%%d, 0, %%%d" % (i % 8, i % 8, (i + 1) % 8),
    "v_mov_b32 sdwa": lambda i: "v_mov_b32_sdwa %%%d, %%%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" % (8 + i % 8, 8 + (i + 3) % 8),
    "v_add_u32 sdwa": lambda i: "v_add_u32_sdwa %%%d, %%16, %%%d dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" % (8 + i % 8, 8 + i % 8),
    "v_mov_b32 dpp row_shr:1": lambda i: "v_mov_b32_dpp %%%d, %%%d row_shr:1 row_mask:0xf bank_mask:0xf" % (8 + i % 8, 8 + (i + 3) % 8),
    "v_dot2_u32_u16": lambda i: "v_dot2_u32_u16 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
    "v_dot4_u32_u8": lambda i: "v_dot4_u32_u8 %%%d, %%16, %%17, %%%d" % (8 + i % 8, 8 + i % 8),
}


def specs():
    out = []
    cls = []
    for name, fn in CLASSIFY.items():
        PLAIN[name] = fn
        cls.append(("%s only" % name, [(name, 32)]))
        cls.append(("addc x1 then %s x1" % name, [("addc", 1), (name, 1)] * 16))
        cls.append(("mad x1 then %s x1" % name, [("mad", 1), (name, 1)] * 16))
    for h in ("mad", "addc"):
        out.append(("%s only" % h, [(h, 32)]))
    for p in ("mov", "add", "xor", "lshl"):
        out.append(("%s only" % p, [(p, 32)]))
    for h in ("mad", "addc", "cnd"):
        for p in ("mov", "add"):
            for k in (1, 2, 4, 8, 16):
                out.append(("%s x%d then %s x%d" % (h, k, p, k), [(h, k), (p, k)] * (16 // k)))
    # the field multiply's real proportions: ~4 heavy per plain
    for p in ("mov", "add"):
        out.append(("(mad addc mad addc %s) x6" % p, ([("mad", 1), ("addc", 1), ("mad", 1), ("addc", 1), (p, 1)]) * 6))
        out.append(("(mad addc)x8 then %s x4" % p, ([("mad", 1), ("addc", 1)] * 8 + [(p, 4)]) * 2))
        out.append(("(mad addc)x16 then %s x8" % p, [("mad", 1), ("addc", 1)] * 16 + [(p, 8)]))
    return out + cls


def main():
    S = specs()
    print("// GENERATED by tools/microbench/gen_issue_runs.py -- do not edit")
    print("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdint>")
    print("typedef uint32_t u32; typedef uint64_t u64;")
    print("template <int P> __global__ void __launch_bounds__(64) k(u32* out, int iters, u32 seed) {")
    print("  u32 a = seed + threadIdx.x * 2654435761u, b = a ^ 0x9E3779B9u;")
    print("  u64 A0 = a, A1 = b, A2 = a + 1, A3 = b + 1, A4 = a + 2, A5 = b + 2, A6 = a + 3, A7 = b + 3;")
    print("  u32 e0 = 1, e1 = 2, e2 = 3, e3 = 4, e4 = 5, e5 = 6, e6 = 7, e7 = 8;")
    print("  for (int it = 0; it < iters; it++) {")
    for idx, (name, spec) in enumerate(S):
        lines, nh, npl = pattern(spec)
        body = "\\n\"\n        \"".join(lines)
        print("    if (P == %d) {  // %s" % (idx, name))
        print("      asm volatile(\"%s\\n\"" % body)
        print("        : \"+v\"(A0), \"+v\"(A1), \"+v\"(A2), \"+v\"(A3), \"+v\"(A4), \"+v\"(A5), \"+v\"(A6), \"+v\"(A7),")
        print("          \"+v\"(e0), \"+v\"(e1), \"+v\"(e2), \"+v\"(e3), \"+v\"(e4), \"+v\"(e5), \"+v\"(e6), \"+v\"(e7)")
        print("        : \"v\"(a), \"v\"(b) : \"s20\", \"s21\", \"s22\", \"s23\", \"s24\", \"s25\");")
        print("    }")
    print("  }")
    print("  u64 s = A0 ^ A1 ^ A2 ^ A3 ^ A4 ^ A5 ^ A6 ^ A7;")
    print("  out[blockIdx.x * 64 + threadIdx.x] = (u32)s ^ (u32)(s >> 32) ^ e0 ^ e1 ^ e2 ^ e3 ^ e4 ^ e5 ^ e6 ^ e7;")
    print("}")
    print("template <int P> static double run(int waves, int iters, u32* out) {")
    print("  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0;")
    print("  for (int rep = 0; rep < 2; rep++) {")
    print("    hipEventRecord(e0); hipLaunchKernelGGL(k<P>, dim3(1024 * waves), dim3(64), 0, 0, out, iters, 12345u);")
    print("    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); }")
    print("  return ms; }")
    print("int main() {")
    print("  u32* out; hipMalloc(&out, 1024 * 16 * 64 * 4); const int iters = 20000;")
    print("  printf(\"ns of SIMD time per wave-instruction at 2, 4, 8 waves per SIMD; [implied ns per PLAIN op, charging the heavy ops their own-run price]\\n\");")
    print("  double heavy_ns[3] = {0, 0, 0};")
    for idx, (name, spec) in enumerate(S):
        lines, nh, npl = pattern(spec)
        print("  { double per[3]; int wi = 0; for (int w : {2, 4, 8}) { double ms = run<%d>(w, iters, out); per[wi++] = ms * 1e6 / ((double)iters * %d * w); }" % (idx, nh + npl))
        print("    printf(\"%%-44s %%6.3f %%6.3f %%6.3f\", \"%s\", per[0], per[1], per[2]);" % name)
        if idx == 0:
            print("    for (int i = 0; i < 3; i++) heavy_ns[i] = per[i];")
        if nh and npl:
            print("    printf(\"   [plain: %%6.3f %%6.3f %%6.3f]\", (per[0] * %d - heavy_ns[0] * %d) / %d, (per[1] * %d - heavy_ns[1] * %d) / %d, (per[2] * %d - heavy_ns[2] * %d) / %d);"
                  % (nh + npl, nh, npl, nh + npl, nh, npl, nh + npl, nh, npl))
        print("    printf(\"\\n\"); }")
    print("  return 0; }")


if __name__ == "__main__":
    main()
