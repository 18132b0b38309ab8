"""Pins oracle/ec_oracle.c (the C port used for bulk checks and the
cpu_baseline) to the reference's golden vectors, and to the Python oracle on
seeded random inputs.  CPU only."""
import random

import numpy as np
import pytest

from elliptic_amd import ints_to_be
from golden_util import I, mul_cases, res_xy, verify_cases
from oracle import c_oracle as C
from oracle import ec_oracle as O

BYTES = {"secp256k1": 32, "p192": 24, "p224": 28, "p256": 32, "p384": 48, "p521": 66}


def _res(out, inf, i, B):
    if inf[i]:
        return None
    return (int.from_bytes(out[i, :B].tobytes(), "big"), int.from_bytes(out[i, B:].tobytes(), "big"))


@pytest.mark.parametrize("name", O.SHORT_CURVES)
def test_c_oracle_mul_golden(name):
    B = BYTES[name]
    cur = O.get_curve(name)
    cases = mul_cases(name)
    fixed = [c for c in cases if c["op"] == "fixed"]
    out, inf = C.mul(name, ints_to_be([I(c["k"]) for c in fixed], B))
    for i, c in enumerate(fixed):
        assert _res(out, inf, i, B) == res_xy(c["r"]), c
    var = [c for c in cases if c["op"] == "var"]
    pts = np.concatenate([ints_to_be([I(c["px"]) for c in var], B), ints_to_be([I(c["py"]) for c in var], B)], axis=1)
    out, inf = C.mul(name, ints_to_be([I(c["k"]) for c in var], B), pts)
    for i, c in enumerate(var):
        assert _res(out, inf, i, B) == res_xy(c["r"]), c
    madd = [c for c in cases if c["op"] == "muladd"]
    k1 = ints_to_be([I(c["k1"]) for c in madd], B)
    k2 = ints_to_be([I(c["k2"]) for c in madd], B)
    p1 = np.concatenate([ints_to_be([I(c["p1x"]) for c in madd], B), ints_to_be([I(c["p1y"]) for c in madd], B)], axis=1)
    p2 = np.concatenate([ints_to_be([I(c["p2x"]) for c in madd], B), ints_to_be([I(c["p2y"]) for c in madd], B)], axis=1)
    out, inf = C.mul_add(name, k1, p1, k2, p2)
    for i, c in enumerate(madd):
        assert _res(out, inf, i, B) == res_xy(c["r"]), c
    gi = [i for i, c in enumerate(madd) if (I(c["p1x"]), I(c["p1y"])) == (cur.g.x, cur.g.y)]
    out, inf = C.mul_add(name, k1[gi], None, k2[gi], p2[gi])
    for j, i in enumerate(gi):
        assert _res(out, inf, j, B) == res_xy(madd[i]["r"]), madd[i]


@pytest.mark.parametrize("name", O.SHORT_CURVES)
def test_c_oracle_verify_golden(name):
    B = BYTES[name]
    groups = {}
    for c in verify_cases(name):
        groups.setdefault((len(c["z"]) // 2, c.get("msgBitLength", 0)), []).append(c)
    n = 0
    for (hl, mb), cs in groups.items():
        ok = C.verify(name, ints_to_be([I(c["z"]) for c in cs], hl), ints_to_be([I(c["r"]) for c in cs], B),
                      ints_to_be([I(c["s"]) for c in cs], B),
                      np.concatenate([ints_to_be([I(c["qx"]) for c in cs], B), ints_to_be([I(c["qy"]) for c in cs], B)], axis=1),
                      msg_bits=mb)
        for i, c in enumerate(cs):
            assert bool(ok[i]) == c["ok"], c
            n += 1
    assert n > 15


def test_c_oracle_vs_python_oracle_random():
    cur = O.get_curve("secp256k1")
    rnd = random.Random(31337)
    n = 200
    ks = [rnd.getrandbits(256) for _ in range(n)]
    ds = [rnd.randrange(1, cur.n) for _ in range(n)]
    pub, inf = C.mul("secp256k1", ints_to_be(ds, 32))
    out, inf = C.mul("secp256k1", ints_to_be(ks, 32), pub)
    for i in range(0, n, 5):
        w = cur.g.mul(ds[i]).mul(ks[i])
        assert _res(out, inf, i, 32) == (None if w.inf else (w.x, w.y))
    ok1 = C.verify("secp256k1", ints_to_be(ks, 32), ints_to_be(ds, 32), ints_to_be(ds[::-1], 32), pub)
    ok8 = C.verify("secp256k1", ints_to_be(ks, 32), ints_to_be(ds, 32), ints_to_be(ds[::-1], 32), pub, threads=4)
    assert np.array_equal(ok1, ok8) and not ok1.any()


def test_c_oracle_edwards_and_montgomery_golden():
    """eco_ed_mul / eco_mont_mul (the bulk checkers of the -m gpu full-size tests) against the
    reference's own results in mul_ed25519.json / mul_curve25519.json"""
    cases = mul_cases("ed25519")
    n = 0
    for op, use_p in (("fixed", False), ("var", True)):
        cs = [c for c in cases if c["op"] == op and I(c["k"]) < (1 << 256)]
        if not cs:
            continue
        ks = ints_to_be([I(c["k"]) for c in cs], 32)
        pts = None
        if use_p:
            pts = np.concatenate([ints_to_be([I(c["px"]) for c in cs], 32), ints_to_be([I(c["py"]) for c in cs], 32)], axis=1)
        out = C.ed_mul(ks, pts, threads=2)
        for i, c in enumerate(cs):
            want = res_xy(c["r"])
            got = (int.from_bytes(out[i, :32].tobytes(), "big"), int.from_bytes(out[i, 32:].tobytes(), "big"))
            assert got == (want if want is not None else (0, 1)), c
            n += 1
    assert n > 20
    cs = [c for c in mul_cases("curve25519") if I(c["k"]) < (1 << 256)]
    out, inf = C.mont_mul(ints_to_be([I(c["k"]) for c in cs], 32), ints_to_be([I(c["px"]) for c in cs], 32))
    m = 0
    for i, c in enumerate(cs):
        r = c["r"]
        if r is None or (isinstance(r, dict) and r.get("inf")):
            assert inf[i] == 1, c
        else:
            x = I(r["x"]) if isinstance(r, dict) else I(r)
            assert inf[i] == 0 and int.from_bytes(out[i].tobytes(), "big") == x, c
        m += 1
    assert m > 5


def test_c_oracle_user_defined_curves_golden():
    """the C restatement's generic-a path (JPoint#_dbl) against the reference's own results on
    user-defined curves (tests/golden/custom_short.json, tools/gen_golden_custom.js)"""
    import parity_checks as PC
    from golden_util import I
    from elliptic_amd import ints_to_be
    n_checked = 0
    for spec in PC.custom_curves():
        if not spec["n"]:
            continue
        name = C.define_short("custom:" + spec["name"], I(spec["p"]), I(spec["a"]), I(spec["b"]), I(spec["n"]),
                              I(spec["g"]["x"]), I(spec["g"]["y"]))
        xy = lambda pts: np.concatenate([ints_to_be([I(q["x"]) for q in pts], 32),
                                         ints_to_be([I(q["y"]) for q in pts], 32)], axis=1)
        want = lambda r: None if r.get("inf") else (I(r["x"]), I(r["y"]))

        def got(out, inf, i):
            return None if inf[i] else (int.from_bytes(out[i, :32].tobytes(), "big"), int.from_bytes(out[i, 32:].tobytes(), "big"))
        mul = [c for c in spec["cases"] if c["op"] == "mul"]
        out, inf = C.mul(name, ints_to_be([I(c["k"]) for c in mul], 32), xy([c["p"] for c in mul]))
        for i, c in enumerate(mul):
            assert got(out, inf, i) == want(c["r"]), (spec["name"], c)
        madd = [c for c in spec["cases"] if c["op"] == "muladd"]
        out, inf = C.mul_add(name, ints_to_be([I(c["k1"]) for c in madd], 32), xy([c["p1"] for c in madd]),
                             ints_to_be([I(c["k2"]) for c in madd], 32), xy([c["p2"] for c in madd]))
        for i, c in enumerate(madd):
            assert got(out, inf, i) == want(c["r"]), (spec["name"], c)
        n_checked += len(mul) + len(madd)
    assert n_checked > 200


@pytest.mark.parametrize("name", O.SHORT_CURVES)
def test_c_oracle_offcurve_golden(name):
    """offcurve_<curve>.json: the reference computes with points that are not on the curve, and the
    C port -- same operation order, same window of G's shipped table -- has to give its results."""
    from golden_util import load
    B = BYTES[name]
    cases = load("offcurve_%s.json" % name)

    def pts(cs, x, y):
        return np.concatenate([ints_to_be([I(c[x]) for c in cs], B), ints_to_be([I(c[y]) for c in cs], B)], axis=1)

    var = [c for c in cases if c["op"] == "var"]
    out, inf = C.mul(name, ints_to_be([I(c["k"]) for c in var], B), pts(var, "px", "py"))
    for i, c in enumerate(var):
        assert _res(out, inf, i, B) == res_xy(c["r"]), c
    for g1 in (True, False):
        madd = [c for c in cases if c["op"] == "muladd" and c["g1"] == g1]
        out, inf = C.mul_add(name, ints_to_be([I(c["k1"]) for c in madd], B), None if g1 else pts(madd, "p1x", "p1y"),
                             ints_to_be([I(c["k2"]) for c in madd], B), pts(madd, "p2x", "p2y"))
        for i, c in enumerate(madd):
            assert _res(out, inf, i, B) == res_xy(c["r"]), c
    ver = [c for c in cases if c["op"] == "verify"]
    hl = len(ver[0]["z"]) // 2
    ok = C.verify(name, ints_to_be([I(c["z"]) for c in ver], hl), ints_to_be([I(c["r"]) for c in ver], B),
                  ints_to_be([I(c["s"]) for c in ver], B), pts(ver, "qx", "qy"))
    for i, c in enumerate(ver):
        assert bool(ok[i]) == c["ok"], c
    assert sum(1 for c in ver if c["ok"] and not c["on"]) >= 8
