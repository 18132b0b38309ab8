#!/usr/bin/env python3
"""Generate elliptic_amd/csrc/k256l_asm.h: gfx950 inline-asm blocks of the 9 x 29-bit signed-limb
multiply / square / two-product multiply of the secp256k1 base field (csrc/fpk256l.h).

Columns of the 17-column product are accumulated in two 64-bit chains with v_mad_i64_i32 and no
carry handling at all (|limb| <= ~2^29: nine products of 2^58 fit 63 bits):

    d-chain   high columns 9..16: h_k = d & M; d >>= 29          -> limbs h_0..h_8 of the high half
    c-chain   low columns 0..8 plus the folded high half (2^261 = R1 * 2^29 + R0 mod p):
              c += column(k) + h_{k-1} * R1 + h_k * R0;  r_k = c & M;  c >>= 29

One asm statement per output limb k (two for the two-product form), the chains interleaved, no
wait states inside a statement.  The low word of a 64-bit accumulator can only be named as an
INPUT operand (`(u32)x` of a value the compiler already holds), so every accumulator is extracted
at the start of the statement after the one that finished it.  The tail (limb 9 and the bits of
limb 8 above 2^24 fold back into limbs 0..2) is plain C++ in fpk256l.h.

    python tools/gen_k256l_asm.py        # writes elliptic_amd/csrc/k256l_asm.h
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "elliptic_amd", "csrc", "k256l_asm.h")
M = "0x1fffffff"


def column(t, sq):
    pr = [(i, t - i) for i in range(9) if 0 <= t - i <= 8]
    return [(i, j) for (i, j) in pr if i <= j] if sq else pr


def statement(name, k, prods_c, prods_d, names, sq, head, fold):
    """one asm statement.
       prods_c / prods_d: products (i, j) added to the c / d chain
       names: (x, y) operand-array names of this product set; sq: squares (x_i * t_j doubled form)
       head: emit the extractions (r_{k-1}, c >>= 29, h = d & M, d >>= 29) first
       fold: emit c += hp*R1 + h*R0 at the end of the c chain"""
    xa, ya = names

    def opnd(i, j):
        if sq:
            return ("%%[%s%d]" % (xa, i), "%%[%s%d]" % (xa, j)) if i == j else ("%%[t%d]" % i, "%%[%s%d]" % (xa, j))
        return ("%%[%s%d]" % (xa, i), "%%[%s%d]" % (ya, j))
    prods = prods_c + prods_d
    ux = sorted({i for (i, j) in prods if (not sq or i == j)} | ({j for (i, j) in prods} if sq else set()))
    ut = sorted({i for (i, j) in prods if i != j}) if sq else []
    uy = sorted({j for (i, j) in prods}) if not sq else []
    decl = ["i32 %s%d" % (xa, i) for i in ux] + ["i32 t%d" % i for i in ut] + ["i32 %s%d" % (ya, j) for j in uy]
    cons = (['[%s%d] "v"(%s%d)' % (xa, i, xa, i) for i in ux] + ['[t%d] "v"(t%d)' % (i, i) for i in ut] +
            ['[%s%d] "v"(%s%d)' % (ya, j, ya, j) for j in uy])
    has_h = head and 0 <= k < 8
    hd, lc, ld = [], [], []
    if head and k > 0:
        hd.append("v_and_b32 %%[r], %s, %%[clo]" % M)
        hd.append("v_ashrrev_i64 %[c], 29, %[ci]")
    if has_h:
        hd.append("v_and_b32 %%[h], %s, %%[dlo]" % M)
        hd.append("v_ashrrev_i64 %[d], 29, %[di]")
    c_in = (k > 0 or not head)           # c has an incoming value
    csrc = ("%[c]" if (head and k > 0) else ("%[ci]" if c_in else "0"))
    for (i, j) in prods_c:
        x, y = opnd(i, j)
        lc.append("v_mad_i64_i32 %%[c], %%[sd], %s, %s, %s" % (x, y, csrc))
        csrc = "%[c]"
    if fold:
        if k > 0:
            lc.append("v_mad_i64_i32 %%[c], %%[sd], %%[hp], %%[R1], %s" % csrc)
            csrc = "%[c]"
        lc.append("v_mad_i64_i32 %%[c], %%[sd], %s, %%[R0], %s" % ("%[h]" if has_h else "%[hin]", csrc))
        csrc = "%[c]"
    d_in = not (k < 0 and head)          # the prologue's first statement starts d at zero
    dsrc = "%[d]" if has_h else ("%[di]" if d_in else "0")
    for (i, j) in prods_d:
        x, y = opnd(i, j)
        ld.append("v_mad_i64_i32 %%[d], %%[sd], %s, %s, %s" % (x, y, dsrc))
        dsrc = "%[d]"
    seq = list(hd)
    a_, b_ = list(lc), list(ld)
    while a_ or b_:
        if b_:
            seq.append(b_.pop(0))
        if a_:
            seq.append(a_.pop(0))
    writes_c = bool(lc) or (head and k > 0)
    writes_d = bool(ld) or has_h
    outs = ['[sd] "=&s"(sd)']
    ins = []
    if writes_c:
        outs.append('[c] "=&v"(co)')
    if writes_d:
        outs.append('[d] "=&v"(do_)')
    if head and k > 0:
        outs.append('[r] "=&v"(ro)')
        ins += ['[clo] "v"((u32)c)', '[ci] "v"(c)']
    elif writes_c and c_in:
        ins.append('[ci] "v"(c)')
    if has_h:
        outs.append('[h] "=&v"(ho)')
        ins += ['[dlo] "v"((u32)d)', '[di] "v"(d)']
    elif writes_d and d_in:
        ins.append('[di] "v"(d)')
    if fold:
        ins.append('[R0] "s"(R0)')
        if k > 0:
            ins += ['[hp] "v"(hp)', '[R1] "s"(R1)']
        if not has_h:
            ins.append('[hin] "v"(h)')
    assert len(outs) + len(ins) + len(cons) <= 30, (name, len(outs) + len(ins) + len(cons))
    args = ["i64& c", "i64& d", "i32& h", "i32 hp", "i32& r", "i32 R0", "i32 R1"]
    code = "ELL_K256L_ASM void %s(%s) {\n  i64 co = 0, do_ = 0; i32 ho = 0, ro = 0; u64 sd;\n" % (name, ", ".join(args + decl))
    code += '  asm("%s"\n      : %s\n      : %s);\n' % ("\\n\\t".join(seq), ", ".join(outs), ", ".join(ins + cons))
    upd = ""
    if writes_c:
        upd += "c = co; "
    if head and k > 0:
        upd += "r = ro; "
    if has_h:
        upd += "h = ho; "
    if writes_d:
        upd += "d = do_; "
    code += "  %s(void)sd; (void)co; (void)do_; (void)ho; (void)ro;\n}\n\n" % upd
    callops = ("".join(", %s[%d]" % (xa, i) for i in ux) + "".join(", t[%d]" % i for i in ut) +
               "".join(", %s[%d]" % (ya, j) for j in uy))
    nm = sum(1 for x in seq if x.startswith("v_mad"))
    return code, "%s(c, d, h, hp, %s, R0, R1%s);" % (name, "r[%d]" % (k - 1) if (head and k > 0) else "dummy", callops), nm


def gen(kind):
    """kind: 'mul' (a*b), 'sqr' (a*a), 'mul2' (a*b + e*f)"""
    sets = {"mul": [("a", "b", False)], "sqr": [("a", "a", True)], "mul2": [("a", "b", False), ("e", "f", False)]}[kind]
    out, calls, mads = "", [], 0
    for k in range(-1, 9):
        for si, (xa, ya, sq) in enumerate(sets):
            first, last = si == 0, si == len(sets) - 1
            pc = column(k, sq) if k >= 0 else []
            pd = column(10 + k, sq) if k <= 6 else []
            if k < 0 and not pd:
                continue
            name = "k256l_%s_S%s_%d" % (kind, "p" if k < 0 else str(k), si)
            code, call, m = statement(name, k, pc, pd, (xa, ya), sq, head=first, fold=(last and k >= 0))
            out += code
            calls.append((k, si, call))
            mads += m
    sig = {"mul": "const i32 (&a)[9], const i32 (&b)[9]", "sqr": "const i32 (&a)[9]",
           "mul2": "const i32 (&a)[9], const i32 (&b)[9], const i32 (&e)[9], const i32 (&f)[9]"}[kind]
    out += "// %d v_mad_i64_i32 in the column statements; returns the carry out of limb 8 (c) and h_8 (hp)\n" % mads
    out += "ELL_K256L_ASM void k256l_%s_cols(i32 (&r)[9], i64& c_out, i32& h8_out, %s) {\n" % (kind, sig)
    out += "  i64 c = 0, d = 0; i32 h = 0, hp = 0, dummy = 0; const i32 R0 = 31264, R1 = 256;\n"
    if kind == "sqr":
        out += "  i32 t[9];\n#pragma unroll\n  for (int i = 0; i < 8; i++) t[i] = a[i] * 2;\n  t[8] = 0;\n"
    for (k, si, call) in calls:
        if si == 0:
            if k == 8:
                out += "  hp = h;\n  h = (i32)d;\n"
            elif k > 0:
                out += "  hp = h;\n"
        out += "  %s\n" % call
    out += "  r[8] = (i32)((u32)c & 0x1fffffffu); c >>= 29;\n  c_out = c; h8_out = h;\n  (void)dummy; (void)hp;\n}\n\n"
    return out


def main():
    hdr = ('// GENERATED by tools/gen_k256l_asm.py -- do not edit.\n'
           '// gfx950 inline asm of the 9 x 29-bit signed-limb products of csrc/fpk256l.h (v_mad_i64_i32\n'
           '// column chains, no carry handling).  Device compilation only.\n'
           '#pragma once\n#include "common.h"\n\n#if defined(__HIP_DEVICE_COMPILE__)\n#define ELL_HAVE_K256L_ASM 1\n'
           '#define ELL_K256L_ASM __device__ __forceinline__\nnamespace ell {\nnamespace k256l {\n'
           'typedef int32_t i32; typedef int64_t i64;\n\n')
    body = gen("mul") + gen("sqr") + gen("mul2")
    tail = "}  // namespace k256l\n}  // namespace ell\n#endif  // __HIP_DEVICE_COMPILE__\n"
    with open(DST, "w") as f:
        f.write(hdr + body + tail)
    print("wrote", DST, len(body.splitlines()), "lines")


if __name__ == "__main__":
    main()
