#!/usr/bin/env python3
"""Static instruction mix of one field multiplication / squaring per curve, from the ISA.

Why: on the NIST curves SQ_INSTS_VALU_INT64 counts the Solinas fold's 64-bit adds and shifts
(v_lshl_add_u64, v_ashrrev_i64 ...) next to the multiplies, so "multiplies issued" cannot be
read off that counter there.  This tool compiles tools/microbench/op_mix.hip for
a curve (hipcc --cuda-device-only -S), counts the instructions on the main path of the mul / sqr
kernels (entry up to the first conditional branch: the rarely taken fold tails sit behind it;
the load/store skeleton of the probe kernel is measured with k_ldst and subtracted) by class --
v_mad_u64_u32, other 64-bit VALU, carry-class, everything else -- and, with the field-operation
counts of the variable-base ladder (a closed formula of the ladder's shape, below), gives the
multiplies one P*k issues.  Checked against PMC where PMC can tell: secp256k1's kernels have no
other 64-bit VALU instruction, and the model lands within 2 % of SQ_INSTS_VALU_INT64 there.

    python tools/static_mix.py [--out profiles/r03_static_op_mix.json]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CURVES = {"secp256k1": "CvSecp256k1", "p192": "CvP192", "p224": "CvP224", "p256": "CvP256", "p384": "CvP384", "p521": "CvP521"}
BYTES = {"secp256k1": 32, "p192": 24, "p224": 28, "p256": 32, "p384": 48, "p521": 66}
INT64_OTHER = re.compile(r"^v_(lshl_add_u64|ashrrev_i64|lshrrev_b64|lshlrev_b64|mad_i64_i32|add_u64|mov_b64)")
CARRY = re.compile(r"^v_(addc|subb|subbrev|add_co|sub_co|subrev_co)")


def classes(lines):
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            continue
        op = l.split()[0]
        if not op.startswith("v_"):
            continue
        if op == "v_mad_u64_u32":
            c["mad_u64_u32"] += 1
        elif INT64_OTHER.match(op):
            c["other_int64"] += 1
        elif CARRY.match(op):
            c["carry"] += 1
        else:
            c["other"] += 1
        c["valu"] += 1
    return c


def probe(curve):
    src = os.path.join(ROOT, "tools", "microbench", "op_mix.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.s")
        subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-w", "-I" + os.path.join(ROOT, "elliptic_amd", "csrc"),
                        "--cuda-device-only", "-S", "-DOPMIX_CURVE=" + CURVES[curve], src, "-o", out], check=True,
                       capture_output=True)
        txt = open(out).read().split("\n")
    res = {}
    for k in ("k_mul", "k_sqr", "k_add", "k_sub", "k_ldst"):
        start = [i for i, l in enumerate(txt) if l.startswith(k + ":")][0]
        end = next(i for i in range(start, len(txt)) if "s_cbranch" in txt[i] or "s_endpgm" in txt[i] or "s_swappc" in txt[i])
        res[k] = classes(txt[start:end])
        res[k]["is_call"] = int("s_swappc" in txt[end])
    base = res.pop("k_ldst")
    out = {}
    for k, c in res.items():
        out[k[2:]] = {f: max(0, c[f] - base[f]) for f in ("valu", "mad_u64_u32", "other_int64", "carry", "other")}
        if c["is_call"]:
            out[k[2:]]["note"] = "the operation is a function call in this build (p521): counts are of the call site only"
    return out


def ladder_field_ops(curve):
    """field multiplications / squarings of ONE variable-base P*k (Work::var_ladder), from the
    ladder's shape: table (doubling, co-Z chain of 7, rescaling of 7), odd-digit ladder"""
    B = BYTES[curve]
    if curve == "secp256k1":
        dbl, madd = (2, 5), (8, 3)                       # a = 0: dbl-2009-l
        windows = 33                                      # GLV halves, two strings
        ndbl, nadd = 4 * (windows - 1), 2 * windows - 1
        extra = (windows + 1, 0)                          # beta * x per lambda*P lookup, Z * zg
        inv = 0
    else:
        dbl, madd = (3, 5), (8, 3)                       # a = -3: dbl-2001-b
        windows = 2 * B + 1
        ndbl, nadd = 4 * (windows - 1), (windows - 1) + 1   # + the "k was even" correction
        extra = (1 + 16, 1)                               # zi3, table mapped back (8 x 2M), zi2
        inv = 1
    M = dbl[0] * (ndbl + 1) + madd[0] * nadd + (1 + 2) + 7 * 4 + 7 * 4 + 1 + extra[0]
    S = dbl[1] * (ndbl + 1) + madd[1] * nadd + 1 + 7 * 2 + 7 * 1 + extra[1]
    return {"mul": M, "sqr": S, "inversions": inv}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--curves", default="secp256k1,p192,p224,p256,p384")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    from elliptic_amd import build as _b
    res = {"source_digest": _b.source_digest(),
           "how": "tools/static_mix.py: main-path instruction classes of one field mul / sqr (ISA, probe skeleton "
                  "subtracted) x the ladder's field-operation counts", "curves": {}}
    for c in a.curves.split(","):
        mix = probe(c)
        ops = ladder_field_ops(c)
        per = {k: ops["mul"] * mix["mul"][k] + ops["sqr"] * mix["sqr"][k] for k in ("mad_u64_u32", "other_int64", "valu")}
        res["curves"][c] = {"per_op": mix, "mul_var_field_ops": ops,
                            "mul_var_model": {"mad_u64_u32_per_unit": per["mad_u64_u32"],
                                              "other_int64_per_unit": per["other_int64"],
                                              "mul_sqr_valu_per_unit": per["valu"],
                                              "note": "multiplications and squarings only (additions, selects, the "
                                                      "division-step inversion of the table's common Z and recoding are not in the model)"}}
        print(c, json.dumps(res["curves"][c]["per_op"]["mul"]), json.dumps(res["curves"][c]["per_op"]["sqr"]), ops, per)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
