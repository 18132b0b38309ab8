#!/usr/bin/env python3
"""Generate elliptic_amd/csrc/mul_asm.h: gfx950 inline-asm multiply-accumulate
blocks for the wide product / square of L-limb integers.

Why asm: the C idiom `(u64)a*b + r + carry` compiles on gfx950 to one
v_mad_u64_u32 plus ~4 v_mov/v_lshl_add_u64 per partial product.  The hardware
has what is needed -- v_mad_u64_u32 D(64) = a*b + C(64) with a carry-out into
an SGPR pair, and v_addc_co_u32 -- but the compiler never uses the carry-out.

Scheme (product scanning, two columns per block): columns k (even) and k+1
each own a 96-bit accumulator (64-bit VGPR pair + 32-bit extension):
    v_mad_u64_u32  acc, sN, a_i, b_j, acc        ; 64-bit accumulate, carry -> sN
    v_addc_co_u32  ext, sD, 0, ext, sN           ; extension += carry
The two columns' chains are independent, and the generator interleaves them so
that every carry is consumed at least 3 issue slots after it is produced (the
VALU-writes-SGPR -> VALU-reads-it-as-carry hazard on gfx950 needs 2 wait
states; nothing pads the inside of an asm statement, so the generator inserts
`s_nop` where interleaving cannot).  Combining the two accumulators of a block
and handing the carry to the next block is left to compiler-visible C++
(__builtin_addc chains, which hipcc pads itself).

    python tools/gen_mul_asm.py        # writes elliptic_amd/csrc/mul_asm.h
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, "elliptic_amd", "csrc", "mul_asm.h")

MIN_DIST = 3          # consumer position - producer position
NSREG = 4             # carry SGPR pairs in flight


def schedule(prods_a, prods_b, a_has_carry_in, b_has_carry_in=False):
    """-> list of instruction tuples:
       ('mad', acc, i, j, sreg|None)   acc in 'A','B'; sreg None => carry impossible, addend literal 0
       ('addc', acc, sreg)
       ('nop',)"""
    pend = []                     # (pos, acc, sreg)
    free = list(range(NSREG))
    seq = []
    qa, qb = list(prods_a), list(prods_b)
    first = {"A": not a_has_carry_in, "B": not b_has_carry_in}   # first product into a zero accumulator cannot carry
    turn = "A"
    while qa or qb or pend:
        pos = len(seq)
        if pend and pos - pend[0][0] >= MIN_DIST:
            _, acc, s = pend.pop(0)
            seq.append(("addc", acc, s))
            free.append(s)
            continue
        # pick a stream: alternate, prefer the longer one when the other is empty
        pick = None
        order = [turn, "B" if turn == "A" else "A"]
        for t in order:
            q = qa if t == "A" else qb
            if q:
                pick = t
                break
        if pick is not None and (free or first[pick]):
            q = qa if pick == "A" else qb
            i, j = q.pop(0)
            if first[pick]:
                seq.append(("mad0", pick, i, j))
                first[pick] = False
            else:
                s = free.pop(0)
                seq.append(("mad", pick, i, j, s))
                pend.append((pos, pick, s))
            turn = "B" if pick == "A" else "A"
            continue
        seq.append(("nop",))
    return seq


def emit_block(name, prods_a, prods_b, a_has_carry_in, square=False, b_has_carry_in=False,
               cont=False):
    """C++ function with one asm statement.

    Every VGPR the statement writes is an EARLY-CLOBBER output ("=&v") distinct from all
    inputs: the statement writes its accumulators long before it has read its last input,
    and with tied "+v" operands LLVM may give an input that holds the same VALUE as an
    accumulator's initial value (e.g. a constant-zero limb and a zero-initialised
    accumulator) the accumulator's own register.  Incoming accumulator values are therefore
    separate read-only inputs (Ai/A2i/Bi/B2i), consumed by the first instruction that
    touches the accumulator.
      cont=False: first statement of a column pair: A comes in (if a_has_carry_in), A2, B, B2
                  start at zero.
      cont=True : continuation statement: all four come in."""
    seq = schedule(prods_a, prods_b, a_has_carry_in or cont, b_has_carry_in or cont)
    used_a = sorted({i for (i, j) in prods_a + prods_b})
    used_b = sorted({j for (i, j) in prods_a + prods_b})
    if square:
        used_a = sorted(set(used_a) | set(used_b))
        used_b = []
    a_in = a_has_carry_in or cont          # A has an incoming value
    a2_in = cont
    b_in = cont
    b2_in = cont
    args = ["u64& A", "u32& A2", "u64& B", "u32& B2"]
    args += ["u32 a%d" % i for i in used_a] + ["u32 b%d" % j for j in used_b]
    lines = []
    touched = {"A": False, "A2": False, "B": False, "B2": False}
    has_in = {"A": a_in, "A2": a2_in, "B": b_in, "B2": b2_in}
    for ins in seq:
        if ins[0] == "nop":
            lines.append("s_nop 0")
        elif ins[0] == "mad0":
            _, acc, i, j = ins
            bj = ("%%[a%d]" % j) if square else ("%%[b%d]" % j)
            assert not has_in[acc] and not touched[acc]
            lines.append("v_mad_u64_u32 %%[%s], %%[sd], %%[a%d], %s, 0" % (acc, i, bj))
            touched[acc] = True
        elif ins[0] == "mad":
            _, acc, i, j, sr = ins
            bj = ("%%[a%d]" % j) if square else ("%%[b%d]" % j)
            src = ("%%[%s]" % acc) if touched[acc] else ("%%[%si]" % acc)
            assert touched[acc] or has_in[acc]
            lines.append("v_mad_u64_u32 %%[%s], %%[s%d], %%[a%d], %s, %s" % (acc, sr, i, bj, src))
            touched[acc] = True
        else:
            _, acc, sr = ins
            e = acc + "2"
            if touched[e]:
                src = "%%[%s]" % e
            elif has_in[e]:
                src = "%%[%si]" % e
            else:
                src = "0"
            lines.append("v_addc_co_u32_e64 %%[%s], %%[sd], 0, %s, %%[s%d]" % (e, src, sr))
            touched[e] = True
    # accumulators this statement never touched still have to be produced
    pre = []
    for acc in ("A", "B"):
        if not touched[acc]:
            pre.append("v_mov_b64 %%[%s], %s" % (acc, ("%%[%si]" % acc) if has_in[acc] else "0"))
            touched[acc] = True
    for e in ("A2", "B2"):
        if not touched[e]:
            pre.append("v_mov_b32 %%[%s], %s" % (e, ("%%[%si]" % e) if has_in[e] else "0"))
            touched[e] = True
    lines = lines + pre          # moves go last: they only copy inputs / constants
    body = "\\n\\t".join(lines)
    outs = ['[A] "=&v"(Ao)', '[A2] "=&v"(A2o)', '[B] "=&v"(Bo)', '[B2] "=&v"(B2o)']
    outs += ['[s%d] "=&s"(s%d)' % (k, k) for k in range(NSREG)] + ['[sd] "=&s"(sd)']
    ins_ = []
    if a_in:
        ins_.append('[Ai] "v"(A)')
    if a2_in:
        ins_.append('[A2i] "v"(A2)')
    if b_in:
        ins_.append('[Bi] "v"(B)')
    if b2_in:
        ins_.append('[B2i] "v"(B2)')
    ins_ += ['[a%d] "v"(a%d)' % (i, i) for i in used_a] + ['[b%d] "v"(b%d)' % (j, j) for j in used_b]
    code = "ELL_DEVASM void %s(%s) {\n" % (name, ", ".join(args))
    code += "  u64 %s, sd, Ao, Bo;\n  u32 A2o, B2o;\n" % ", ".join("s%d" % k for k in range(NSREG))
    code += '  asm("%s"\n      : %s\n      : %s\n      : );\n' % (body, ", ".join(outs), ", ".join(ins_))
    code += "  A = Ao; A2 = A2o; B = Bo; B2 = B2o;\n"
    code += "  (void)sd;" + "".join(" (void)s%d;" % k for k in range(NSREG)) + "\n}\n\n"
    nmad = sum(1 for x in seq if x[0] in ("mad", "mad0"))
    nadd = sum(1 for x in seq if x[0] == "addc")
    nnop = sum(1 for x in seq if x[0] == "nop")
    return code, used_a, used_b, (nmad, nadd, nnop)


CHUNK = 8   # distinct a-limbs per asm statement (inline asm allows 30 operands)


def emit_chunked(base, pa, pb, a_cin, square=False):
    """split a column pair's products into asm statements of <= CHUNK a-limbs each;
    returns (code, [(name, used_a, used_b)], stats)"""
    idx = sorted({i for (i, j) in pa + pb})
    code = ""
    calls = []
    stats = [0, 0, 0]
    a_in, b_in = a_cin, False
    for c in range(0, max(len(idx), 1), CHUNK):
        sel = set(idx[c:c + CHUNK])
        ca = [(i, j) for (i, j) in pa if i in sel]
        cb = [(i, j) for (i, j) in pb if i in sel]
        if not ca and not cb:
            continue
        name = "%s_%d" % (base, c // CHUNK)
        cd, ua, ub, st = emit_block(name, ca, cb, a_cin if c == 0 else True, square, False, cont=(c > 0))
        code += cd
        calls.append((name, ua, ub))
        for t in range(3):
            stats[t] += st[t]
        if ca:
            a_in = True
        if cb:
            b_in = True
    return code, calls, stats


def gen_mul(L):
    """full product of two L-limb numbers -> 2L limbs"""
    out = []
    calls = []
    stats = [0, 0, 0]
    for k in range(0, 2 * L - 1, 2):
        pa = [(i, k - i) for i in range(L) if 0 <= k - i < L]
        pb = [(i, k + 1 - i) for i in range(L) if 0 <= k + 1 - i < L]
        code, cl, st = emit_chunked("mulblk%d_%d" % (L, k), pa, pb, k > 0)
        out.append(code)
        calls.append((k, cl))
        for t in range(3):
            stats[t] += st[t]
    fn = "// r[0..%d) = a * b   (%d v_mad_u64_u32, %d v_addc, %d s_nop in asm + C++ combines)\n" % (2 * L, *stats)
    fn += "ELL_DEVASM void mul_wide_%d(u32 (&r)[%d], const u32 (&a)[%d], const u32 (&b)[%d]) {\n" % (L, 2 * L, L, L)
    fn += "  u64 A = 0; u32 A2 = 0;\n"
    for (k, cl) in calls:
        fn += "  { u64 B = 0; u32 B2 = 0;\n"
        for (name, ua, ub) in cl:
            fn += "    %s(A, A2, B, B2%s%s);\n" % (name, "".join(", a[%d]" % i for i in ua), "".join(", b[%d]" % j for j in ub))
        fn += "    r[%d] = (u32)A;\n" % k
        fn += "    u32 c; u32 t1 = addc32((u32)(A >> 32), (u32)B, 0, c);\n"
        fn += "    u32 n0 = addc32(A2, (u32)(B >> 32), c, c);\n"
        fn += "    u32 n1 = B2 + c;\n"
        fn += "    r[%d] = t1;\n" % (k + 1)
        fn += "    A = ((u64)n1 << 32) | n0; A2 = 0; }\n"
    fn += "}\n\n"
    return "".join(out) + fn


def gen_sqr_off(L):
    """off-diagonal half: sum_{i<j} a_i a_j 2^(32(i+j)) -> 2L limbs (top limb 0)"""
    out = []
    calls = []
    stats = [0, 0, 0]
    for k in range(0, 2 * L - 1, 2):
        pa = [(i, k - i) for i in range(L) if 0 <= k - i < L and i < k - i]
        pb = [(i, k + 1 - i) for i in range(L) if 0 <= k + 1 - i < L and i < k + 1 - i]
        if not pa and not pb:
            calls.append((k, []))
            continue
        code, cl, st = emit_chunked("sqrblk%d_%d" % (L, k), pa, pb, k > 0, square=True)
        out.append(code)
        calls.append((k, cl))
        for t in range(3):
            stats[t] += st[t]
    fn = "// r[0..%d) = sum_{i<j} a_i*a_j*2^(32(i+j))   (%d v_mad_u64_u32, %d v_addc, %d s_nop)\n" % (2 * L, *stats)
    fn += "ELL_DEVASM void sqr_offdiag_%d(u32 (&r)[%d], const u32 (&a)[%d]) {\n" % (L, 2 * L, L)
    fn += "  u64 A = 0; u32 A2 = 0;\n"
    for (k, cl) in calls:
        fn += "  { u64 B = 0; u32 B2 = 0;\n"
        for (name, ua, ub) in cl:
            fn += "    %s(A, A2, B, B2%s);\n" % (name, "".join(", a[%d]" % i for i in ua))
        fn += "    r[%d] = (u32)A;\n" % k
        fn += "    u32 c; u32 t1 = addc32((u32)(A >> 32), (u32)B, 0, c);\n"
        fn += "    u32 n0 = addc32(A2, (u32)(B >> 32), c, c);\n"
        fn += "    u32 n1 = B2 + c;\n"
        fn += "    r[%d] = t1;\n" % (k + 1)
        fn += "    A = ((u64)n1 << 32) | n0; A2 = 0; }\n"
    fn += "}\n\n"
    return "".join(out) + fn


# --------------------------------------------------------------------------------------------
# Single-chain variant: one asm statement per column (or per <= CHAIN_CH products of a column),
# ONE 64-bit accumulator X + 32-bit extension E.  The next column starts from
# Xin = (X.hi, E): the pair is assembled by the compiler between the statements (one v_mov),
# which replaces the three carry-consuming adds of the two-column combine.  On gfx950 every
# carry-consuming instruction issues at the cost of a v_mad_u64_u32, a plain v_mov at ~0.6 of it.
# --------------------------------------------------------------------------------------------
CHAIN_CH = 10       # products per statement (30-operand limit)


def chain_stmt(name, prods, has_xin, has_ein, may_carry_first, need_e, square=False, ovf_first=False):
    """one statement: X = Xin + sum(products), E = Ein + carries.
       has_xin/has_ein: incoming values; may_carry_first: the first mad may overflow
       need_e: the statement has to produce E at all
       ovf_first: the first mad's carry-out (Xin = (X.hi, E) of the previous column is < 2^36, so
       it overflows only when a_i * b_j >= 2^64 - 2^36: never for random operands) is not added
       to E but OR-ed into the caller's overflow mask on the scalar unit; the caller redoes the
       product with the all-carries variant when the mask is non-zero"""
    # schedule: mads in order; each carrying mad's addc issued >= MIN_DIST positions later
    seq = []
    pend = []
    free = list(range(NSREG))
    q = list(prods)
    first = True
    while q or pend:
        pos = len(seq)
        # multiplies first while a carry register is free (puts distance between a carry and
        # its consumer), then the oldest pending carry, then a wait state
        if q:
            carry = need_e and (may_carry_first or not first)
            if carry and first and ovf_first:
                i, j = q.pop(0)
                seq.append(("madf", i, j))
                first = False
                continue
            if not carry:
                i, j = q.pop(0)
                seq.append(("madnc", i, j))
                first = False
                continue
            if free:
                i, j = q.pop(0)
                sr = free.pop(0)
                seq.append(("mad", i, j, sr))
                pend.append((pos, sr))
                first = False
                continue
        if pend and pos - pend[0][0] >= MIN_DIST:
            _, sr = pend.pop(0)
            seq.append(("addc", sr))
            free.append(sr)
            continue
        seq.append(("nop",))
    used_a = sorted({i for (i, j) in prods})
    used_b = sorted({j for (i, j) in prods})
    if square:
        used_a = sorted(set(used_a) | set(used_b))
        used_b = []
    lines = []
    x_touched = False
    e_touched = False
    for ins in seq:
        if ins[0] == "nop":
            lines.append("s_nop 0")
            continue
        if ins[0] in ("mad", "madnc", "madf"):
            i, j = ins[1], ins[2]
            bj = ("%%[a%d]" % j) if square else ("%%[b%d]" % j)
            src = "%[X]" if x_touched else ("%[Xi]" if has_xin else "0")
            sr = "%%[s%d]" % ins[3] if ins[0] == "mad" else ("%[sf]" if ins[0] == "madf" else "%[sd]")
            lines.append("v_mad_u64_u32 %%[X], %s, %%[a%d], %s, %s" % (sr, i, bj, src))
            x_touched = True
        else:
            src = "%[E]" if e_touched else ("%[Ei]" if has_ein else "0")
            lines.append("v_addc_co_u32_e64 %%[E], %%[sd], 0, %s, %%[s%d]" % (src, ins[1]))
            e_touched = True
    if need_e and not e_touched:
        lines.append("v_mov_b32 %%[E], %s" % ("%[Ei]" if has_ein else "0"))
    has_f = any(x[0] == "madf" for x in seq)
    if has_f:
        # the scalar OR goes last: at least two instructions behind the mad that wrote sf
        while len(lines) < 3:
            lines.append("s_nop 0")
        lines.append("s_or_b64 %[ovf], %[ovf], %[sf]")
    body = "\\n\\t".join(lines)
    args = ["u64& X"] + (["u32& E"] if need_e else []) + (["u64& ovf"] if has_f else [])
    args += ["u32 a%d" % i for i in used_a] + ["u32 b%d" % j for j in used_b]
    outs = ['[X] "=&v"(Xo)'] + (['[E] "=&v"(Eo)'] if need_e else [])
    outs += ['[s%d] "=&s"(s%d)' % (k, k) for k in range(NSREG)] + ['[sd] "=&s"(sd)']
    if has_f:
        outs += ['[sf] "=&s"(sf)', '[ovf] "+&s"(ovf)']
    ins_ = []
    if has_xin:
        ins_.append('[Xi] "v"(X)')
    if has_ein:
        ins_.append('[Ei] "v"(E)')
    ins_ += ['[a%d] "v"(a%d)' % (i, i) for i in used_a] + ['[b%d] "v"(b%d)' % (j, j) for j in used_b]
    code = "ELL_DEVASM void %s(%s) {\n" % (name, ", ".join(args))
    code += "  u64 %s, sd, Xo;\n" % ", ".join("s%d" % k for k in range(NSREG))
    if has_f:
        code += "  u64 sf;\n"
    if need_e:
        code += "  u32 Eo;\n"
    code += '  asm("%s"\n      : %s\n      : %s\n      : %s);\n' % (body, ", ".join(outs), ", ".join(ins_) if ins_ else "",
                                                               '"scc"' if has_f else "")
    code += "  X = Xo;" + (" E = Eo;" if need_e else "") + "\n"
    code += "  (void)sd;" + ("(void)sf;" if has_f else "") + "".join(" (void)s%d;" % k for k in range(NSREG)) + "\n}\n\n"
    st = (sum(1 for x in seq if x[0] in ("mad", "madnc", "madf")), sum(1 for x in seq if x[0] == "addc"),
          sum(1 for x in seq if x[0] == "nop"))
    return code, used_a, used_b, st, has_f


def gen_chain(L, square, fast=False):
    """single-chain wide product (square=False) or off-diagonal half of the square.
    fast=True: the variant whose columns do not add their first multiply-add's carry-out to E
    (see chain_stmt, ovf_first); it takes a u64& ovf that comes back non-zero when that was
    wrong for some lane, and the caller then calls the plain variant."""
    out = []
    stats = [0, 0, 0]
    base = (("sqrf%d" if square else "mulf%d") if fast else ("sqrc%d" if square else "mulc%d")) % L
    cols = []
    for k in range(0, 2 * L - 1):
        pr = [(i, k - i) for i in range(L) if 0 <= k - i < L and (not square or i < k - i)]
        cols.append(pr)
    nz = [k for k in range(len(cols)) if cols[k]]
    first_col, last_col = nz[0], nz[-1]
    fn_name = ((("sqrf_offdiag_%d" if square else "mulf_wide_%d") if fast else
                ("sqrc_offdiag_%d" if square else "mulc_wide_%d"))) % L
    ovf_arg = ", u64& ovf" if fast else ""
    if square:
        fn = "// r[0..%d) = sum_{i<j} a_i*a_j*2^(32(i+j)), single accumulator chain\n" % (2 * L)
        fn += "ELL_DEVASM void %s(u32 (&r)[%d], const u32 (&a)[%d]%s) {\n" % (fn_name, 2 * L, L, ovf_arg)
    else:
        fn = "// r[0..%d) = a * b, single accumulator chain\n" % (2 * L)
        fn += "ELL_DEVASM void %s(u32 (&r)[%d], const u32 (&a)[%d], const u32 (&b)[%d]%s) {\n" % (fn_name, 2 * L, L, L, ovf_arg)
    fn += "  u64 X = 0; u32 E = 0;\n"
    for k in range(first_col):
        fn += "  r[%d] = 0;\n" % k
    e_prev_zero = True          # E of the previous column is known to be zero
    for k in range(first_col, last_col + 1):
        pr = cols[k]
        is_first = k == first_col
        is_last = k == last_col
        # the last column's sum cannot overflow 64 bits (the whole value fits 2L limbs); a single
        # product on top of Xin < 2^32 cannot either
        need_e = not is_last and not (len(pr) == 1 and e_prev_zero)
        chunks = [pr[c:c + CHAIN_CH] for c in range(0, len(pr), CHAIN_CH)]
        for ci, ch in enumerate(chunks):
            name = "%s_%d_%d" % (base, k, ci)
            has_xin = (not is_first) or ci > 0
            has_ein = ci > 0 and need_e
            # first mad of the column: Xin = (X.hi, E_prev) < 2^32 when E_prev == 0 -> cannot overflow
            may_carry_first = (ci > 0) or (not e_prev_zero)
            code, ua, ub, st, has_f = chain_stmt(name, ch, has_xin, has_ein, may_carry_first, need_e, square,
                                                 ovf_first=fast and ci == 0 and has_xin)
            out.append(code)
            for t in range(3):
                stats[t] += st[t]
            call_args = ["X"] + (["E"] if need_e else []) + (["ovf"] if has_f else [])
            call_args += ["a[%d]" % i for i in ua] + ["b[%d]" % j for j in ub]
            fn += "  %s(%s);\n" % (name, ", ".join(call_args))
        fn += "  r[%d] = (u32)X;\n" % k
        if not is_last:
            if need_e:
                fn += "  X = ((u64)E << 32) | (X >> 32);\n"
                e_prev_zero = False
            else:
                fn += "  X = X >> 32;\n"
        else:
            fn += "  r[%d] = (u32)(X >> 32);\n" % (k + 1)
    for k in range(last_col + 2, 2 * L):
        fn += "  r[%d] = 0;\n" % k
    fn += "}\n"
    fn = fn.replace("single accumulator chain", "single accumulator chain (%d v_mad_u64_u32, %d v_addc, %d s_nop)" % tuple(stats))
    return "".join(out) + fn + "\n"


def gen_fold4():
    """x_i = h_i * k + P_i for four (h, P) pairs, P a 64-bit addend; returns the OR of the four
    carry-out masks (non-zero when any lane overflowed 64 bits in any of them)"""
    body = "\\n\\t".join(
        ["v_mad_u64_u32 %%[x%d], %%[c%d], %%[h%d], %%[k], %%[P%d]" % (i, i, i, i) for i in range(4)] +
        ["s_nop 1",
         "s_or_b64 %[m], %[c0], %[c1]",
         "s_or_b64 %[c2], %[c2], %[c3]",
         "s_or_b64 %[m], %[m], %[c2]"])
    code = "// x_i = h_i * k + P_i (i = 0..3); returns the OR of the carry-out lane masks\n"
    code += "ELL_DEVASM u64 fold4(u64& x0, u64& x1, u64& x2, u64& x3, u64 P0, u64 P1, u64 P2, u64 P3,\n"
    code += "                     u32 h0, u32 h1, u32 h2, u32 h3, u32 k) {\n"
    code += "  u64 c0, c1, c2, c3, m, o0, o1, o2, o3;\n"
    code += '  asm("%s"\n' % body
    code += '      : [x0] "=&v"(o0), [x1] "=&v"(o1), [x2] "=&v"(o2), [x3] "=&v"(o3), [m] "=&s"(m),\n'
    code += '        [c0] "=&s"(c0), [c1] "=&s"(c1), [c2] "=&s"(c2), [c3] "=&s"(c3)\n'
    code += '      : [P0] "v"(P0), [P1] "v"(P1), [P2] "v"(P2), [P3] "v"(P3),\n'
    code += '        [h0] "v"(h0), [h1] "v"(h1), [h2] "v"(h2), [h3] "v"(h3), [k] "s"(k)\n'
    code += '      : "scc");\n'
    code += "  x0 = o0; x1 = o1; x2 = o2; x3 = o3;\n"
    code += "  (void)c0; (void)c1; (void)c2; (void)c3;\n  return m;\n}\n\n"
    return code


def main():
    hdr = ('// GENERATED by tools/gen_mul_asm.py -- do not edit.\n'
           '// gfx950 inline-asm multiply-accumulate blocks (v_mad_u64_u32 with carry-out into an\n'
           '// SGPR pair + v_addc_co_u32), hazard-spaced by the generator.  Device compilation only;\n'
           '// host passes (hipcc host side, tests/hostsim) use the portable code in common.h.\n'
           '#pragma once\n#include "common.h"\n\n'
           '#if defined(__HIP_DEVICE_COMPILE__)\n'
           '#define ELL_HAVE_MUL_ASM 1\n'
           '#define ELL_DEVASM __device__ __forceinline__\n'
           'namespace ell {\nnamespace masm {\n\n')
    body = ""
    for L in (6, 7, 8, 12, 17):
        body += gen_mul(L)
        body += gen_sqr_off(L)
        body += gen_chain(L, False)
        body += gen_chain(L, True)
        body += gen_chain(L, False, fast=True)
        body += gen_chain(L, True, fast=True)
    body += gen_fold4()
    tail = "}  // namespace masm\n}  // namespace ell\n#endif  // __HIP_DEVICE_COMPILE__\n"
    with open(DST, "w") as f:
        f.write(hdr + body + tail)
    print("wrote", DST, len(body.splitlines()), "lines")


if __name__ == "__main__":
    main()
