"""Helpers to read tests/golden/*.json (made by tools/gen_golden.js from the
reference bundle)."""
import json
import os

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def I(s):
    return int(s, 16)


def res_xy(r):
    """golden result -> None (infinity) or (x, y) / (x,) ints."""
    if r.get("inf"):
        return None
    if "y" in r:
        return (I(r["x"]), I(r["y"]))
    return (I(r["x"]),)


def mul_cases(curve):
    """seeded cases + the calls captured from the reference's own suite."""
    cases = list(load("mul_%s.json" % curve))
    cap = load("captured_%s.json" % curve)
    for c in cap["mul"]:
        c = dict(c)
        c["op"] = "ladder" if curve == "curve25519" else "var"
        c["captured"] = True
        cases.append(c)
    for c in cap["muladd"]:
        c = dict(c)
        c["op"] = "muladd"
        c["captured"] = True
        cases.append(c)
    return cases


def verify_cases(curve):
    cases = list(load("verify_%s.json" % curve))
    for c in load("captured_%s.json" % curve)["verify"]:
        c = dict(c)
        c["captured"] = True
        cases.append(c)
    return cases
