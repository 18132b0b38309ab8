"""ctypes wrapper of oracle/ec_oracle.c (the C port of the reference's
algorithm).  TEST INFRASTRUCTURE ONLY -- imported by tests/, smoke() and
bench.py's cpu_baseline leg, never by the product."""
import ctypes
import json
import os
import subprocess
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libec_oracle.so")
_GOLDEN = os.path.join(HERE, "..", "tests", "golden", "curves.json")


def build(force=False):
    src = os.path.join(HERE, "ec_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.run(["make", "-C", HERE, "-B" if force else "-s"], check=True, capture_output=True)
    return LIB


_lib = None
_curves = {}


def load():
    global _lib
    if _lib is None:
        build()
        lib = ctypes.CDLL(LIB)
        vp = ctypes.c_void_p
        lib.eco_curve_create.restype = vp
        lib.eco_curve_create.argtypes = [ctypes.c_int, ctypes.c_int] + [vp] * 6 + [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        lib.eco_mul.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp]
        lib.eco_mul_add.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp, vp, vp]
        lib.eco_verify.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
        lib.eco_verify_mt.argtypes = [vp, ctypes.c_size_t, vp, ctypes.c_int, ctypes.c_int, vp, vp, vp, vp, ctypes.c_int]
        lib.eco_ed_create.restype = vp
        lib.eco_ed_create.argtypes = [ctypes.c_int, ctypes.c_int] + [vp] * 5
        lib.eco_ed_mul.argtypes = [vp, ctypes.c_size_t, vp, vp, vp]
        lib.eco_mont_create.restype = vp
        lib.eco_mont_create.argtypes = [ctypes.c_int, vp, vp]
        lib.eco_mont_mul.argtypes = [vp, ctypes.c_size_t, vp, vp, vp, vp]
        lib.eco_bulk_mt.argtypes = [ctypes.c_int, vp, ctypes.c_size_t, vp, vp, vp, vp, ctypes.c_int]
        _lib = lib
    return _lib


def _signed_hex(s):
    return (-int(s[1:], 16), True) if s.startswith("-") else (int(s, 16), False)


def curve(name):
    """handle of a short-Weierstrass preset (parameters from the fixture the
    reference dumped, tests/golden/curves.json)"""
    if name in _curves:
        return _curves[name]
    lib = load()
    with open(_GOLDEN) as f:
        c = json.load(f)[name]
    assert c["type"] == "short"
    B = c["bytes"]
    NB = len(c["n"]) // 2
    b = lambda h, w: (ctypes.c_char * w).from_buffer_copy(int(h, 16).to_bytes(w, "big"))
    bufs = [b(c[k], B) for k in ("p", "a", "b")] + [b(c["n"], NB), b(c["gx"], B), b(c["gy"], B)]
    endo = 1 if "beta" in c else 0
    beta = lam = basis = None
    negs = (ctypes.c_int * 4)(0, 0, 0, 0)
    keep = list(bufs)
    if endo:
        beta, lam = b(c["beta"], B), b(c["lambda"], NB)
        raw = b""
        for i, v in enumerate(c["basis"]):
            for j, key in enumerate(("a", "b")):
                val, neg = _signed_hex(v[key])
                raw += abs(val).to_bytes(17, "big")
                negs[2 * i + j] = 1 if neg else 0
        basis = (ctypes.c_char * len(raw)).from_buffer_copy(raw)
        keep += [beta, lam, basis]
    # secp256k1 ships naf wnd 7 tables (precomputed/secp256k1.js); the others get
    # precompute()'s wnd 8 (base.js:312-327)
    wnd = 7 if name == "secp256k1" else 8
    addr = lambda x: ctypes.addressof(x) if x is not None else None
    h = lib.eco_curve_create(B, NB, *[addr(x) for x in bufs], wnd, endo, addr(beta), addr(lam), addr(basis),
                             ctypes.addressof(negs))
    _curves[name] = (h, B, NB, keep)
    return _curves[name]


def define_short(name, p, a, b, n, gx, gy):
    """register a user-defined short curve (arbitrary a: JPoint#_dbl, short.js:802-830) under
    `name`, 32-byte widths like ellgpu_curve_define_short; usable with mul / mul_mt / mul_add"""
    if name in _curves:
        return name
    lib = load()
    b32 = lambda v: (ctypes.c_char * 32).from_buffer_copy(int(v).to_bytes(32, "big"))
    bufs = [b32(p), b32(a % p), b32(b % p), b32(n), b32(gx), b32(gy)]
    negs = (ctypes.c_int * 4)(0, 0, 0, 0)
    h = lib.eco_curve_create(32, 32, *[ctypes.addressof(x) for x in bufs], 8, 0, None, None, None,
                             ctypes.addressof(negs))
    _curves[name] = (h, 32, 32, bufs)
    return name


def _p(a):
    return None if a is None else a.ctypes.data


_special = {}


def _special_curve(name):
    """handle of ed25519 (Edwards) or curve25519 (Montgomery), parameters from curves.json"""
    if name in _special:
        return _special[name]
    lib = load()
    with open(_GOLDEN) as f:
        c = json.load(f)[name]
    B = c["bytes"]
    b = lambda h: (ctypes.c_char * B).from_buffer_copy(int(h, 16).to_bytes(B, "big"))
    if c["type"] == "edwards":
        bufs = [b(c[k]) for k in ("p", "a", "d", "gx", "gy")]
        h = lib.eco_ed_create(B, c["nbits"], *[ctypes.addressof(x) for x in bufs])
    else:
        bufs = [b(c[k]) for k in ("p", "a")]
        h = lib.eco_mont_create(B, *[ctypes.addressof(x) for x in bufs])
    _special[name] = (h, B, bufs)
    return _special[name]


def ed_mul(k, xy=None, threads=1, name="ed25519"):
    """k*P on ed25519 (xy None: the generator, with Point#mul's table dispatch) -> affine x||y"""
    h, B, _ = _special_curve(name)
    k = np.ascontiguousarray(k, np.uint8).reshape(-1, B)
    n = k.shape[0]
    if xy is not None:
        xy = np.ascontiguousarray(xy, np.uint8).reshape(n, 2 * B)
    out = np.zeros((n, 2 * B), np.uint8)
    load().eco_bulk_mt(1, h, n, _p(k), _p(xy), _p(out), None, max(1, threads))
    return out


def mont_mul(k, x, threads=1, name="curve25519"):
    """getX(k * (x : 1)) on curve25519 -> (x bytes, inf flags)"""
    h, B, _ = _special_curve(name)
    k = np.ascontiguousarray(k, np.uint8).reshape(-1, B)
    n = k.shape[0]
    x = np.ascontiguousarray(x, np.uint8).reshape(n, B)
    out = np.zeros((n, B), np.uint8)
    inf = np.zeros(n, np.uint8)
    load().eco_bulk_mt(2, h, n, _p(k), _p(x), _p(out), _p(inf), max(1, threads))
    return out, inf


def mul_mt(name, k, xy=None, threads=1):
    """mul() over `threads` slices"""
    h, B, NB, _ = curve(name)
    k = np.ascontiguousarray(k, np.uint8).reshape(-1, B)
    n = k.shape[0]
    if xy is not None:
        xy = np.ascontiguousarray(xy, np.uint8).reshape(n, 2 * B)
    out = np.zeros((n, 2 * B), np.uint8)
    inf = np.zeros(n, np.uint8)
    load().eco_bulk_mt(0, h, n, _p(k), _p(xy), _p(out), _p(inf), max(1, threads))
    return out, inf


def mul(name, k, xy=None):
    h, B, NB, _ = curve(name)
    k = np.ascontiguousarray(k, np.uint8).reshape(-1, B)
    n = k.shape[0]
    if xy is not None:
        xy = np.ascontiguousarray(xy, np.uint8).reshape(n, 2 * B)
    out = np.zeros((n, 2 * B), np.uint8)
    inf = np.zeros(n, np.uint8)
    load().eco_mul(h, n, _p(k), _p(xy), _p(out), _p(inf))
    return out, inf


def mul_add(name, k1, xy1, k2, xy2):
    h, B, NB, _ = curve(name)
    k1 = np.ascontiguousarray(k1, np.uint8).reshape(-1, B)
    n = k1.shape[0]
    k2 = np.ascontiguousarray(k2, np.uint8).reshape(n, B)
    xy2 = np.ascontiguousarray(xy2, np.uint8).reshape(n, 2 * B)
    if xy1 is not None:
        xy1 = np.ascontiguousarray(xy1, np.uint8).reshape(n, 2 * B)
    out = np.zeros((n, 2 * B), np.uint8)
    inf = np.zeros(n, np.uint8)
    load().eco_mul_add(h, n, _p(k1), _p(xy1), _p(k2), _p(xy2), _p(out), _p(inf))
    return out, inf


def verify(name, hashes, r, s, pub, msg_bits=0, threads=1):
    h, B, NB, _ = curve(name)
    hashes = np.ascontiguousarray(hashes, np.uint8)
    n, hl = hashes.shape
    r = np.ascontiguousarray(r, np.uint8).reshape(n, NB)
    s = np.ascontiguousarray(s, np.uint8).reshape(n, NB)
    pub = np.ascontiguousarray(pub, np.uint8).reshape(n, 2 * B)
    ok = np.zeros(n, np.uint8)
    if threads > 1:
        load().eco_verify_mt(h, n, _p(hashes), hl, msg_bits, _p(r), _p(s), _p(pub), _p(ok), threads)
    else:
        load().eco_verify(h, n, _p(hashes), hl, msg_bits, _p(r), _p(s), _p(pub), _p(ok))
    return ok


def _usable_cpus():
    """threads this process may actually use: affinity mask, capped by a cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def bench_verify(lib, h, r, s, pub, expect, budget_s=15.0):
    """cpu_baseline leg of bench.py: time the port on a bounded sample of the
    batch -- 1 thread, then a few thread counts up to the usable CPUs -- and report
    the best aggregate; every run is checked against the expected mask."""
    m1 = 2000
    t0 = time.perf_counter()
    ok = verify("secp256k1", h[:m1], r[:m1], s[:m1], pub[:m1])
    dt1 = time.perf_counter() - t0
    assert np.array_equal(ok, expect[:m1]), "C oracle disagrees with the expected mask"
    rate1 = m1 / dt1
    usable = _usable_cpus()
    tried = {}
    best = (rate1, 1, m1)
    spent = dt1
    for t in sorted({4, 16, 64, usable, os.cpu_count() or 1}):
        if t <= 1 or spent > budget_s:
            continue
        m = int(min(len(expect), max(m1, rate1 * min(t, usable) * 1.5)))
        t0 = time.perf_counter()
        ok = verify("secp256k1", h[:m], r[:m], s[:m], pub[:m], threads=t)
        dt = time.perf_counter() - t0
        spent += dt
        assert np.array_equal(ok, expect[:m]), "C oracle disagrees with the expected mask"
        tried[t] = m / dt
        if m / dt > best[0]:
            best = (m / dt, t, m)
    return {"value": best[0], "unit": "verifies/s", "cores": best[1], "kind": "port",
            "single_core_value": rate1, "usable_cpus": usable, "host_cpus": os.cpu_count(),
            "threads_tried": {str(k): v for k, v in tried.items()},
            "sample": "first %d tuples of the rank-0 batch on %d threads (best of the thread counts tried; "
                      "the first %d on 1 thread), oracle/ec_oracle.c = C port of the reference's "
                      "wNAF/JSF/GLV ladder" % (best[2], best[1], m1)}
