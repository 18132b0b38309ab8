"""Parity checks shared by the CPU (hostsim) and GPU test modules: drive a
Context through the C ABI and compare with the golden fixtures (reference
outputs) and with the oracle."""
import numpy as np

from elliptic_amd import FIELD_BYTES, ORDER_BYTES, be_to_ints, ints_to_be
from golden_util import I, mul_cases, res_xy, verify_cases


def _res_from(out, inf, i, B, xonly=False):
    if inf[i]:
        return None
    x = int.from_bytes(out[i, :B].tobytes(), "big")
    if xonly:
        return (x,)
    y = int.from_bytes(out[i, B:2 * B].tobytes(), "big")
    return (x, y)


def check_mul_golden(ctx, curve):
    """every golden mul/muladd case of `curve` (seeded, edge, and captured from
    the reference's own test-suite) through the batched C ABI."""
    B = FIELD_BYTES[curve]
    cases = mul_cases(curve)
    fixed = [c for c in cases if c["op"] == "fixed"]
    var = [c for c in cases if c["op"] == "var"]
    madd = [c for c in cases if c["op"] == "muladd"]
    n_checked = 0
    if fixed:
        out, inf = ctx.mul_fixed(curve, ints_to_be([I(c["k"]) for c in fixed], B))
        for i, c in enumerate(fixed):
            want = res_xy(c["r"])
            got = _res_from(out, inf, i, B)
            assert got == want, ("fixed", curve, c)
            n_checked += 1
    if var:
        ks = ints_to_be([I(c["k"]) for c in var], B)
        pts = np.concatenate([ints_to_be([I(c["px"]) for c in var], B),
                              ints_to_be([I(c["py"]) for c in var], B)], axis=1)
        out, inf = ctx.mul_var(curve, ks, pts)
        for i, c in enumerate(var):
            assert _res_from(out, inf, i, B) == res_xy(c["r"]), ("var", curve, c)
            n_checked += 1
    if madd:
        k1 = ints_to_be([I(c["k1"]) for c in madd], B)
        k2 = ints_to_be([I(c["k2"]) for c in madd], B)
        p1 = np.concatenate([ints_to_be([I(c["p1x"]) for c in madd], B),
                             ints_to_be([I(c["p1y"]) for c in madd], B)], axis=1)
        p2 = np.concatenate([ints_to_be([I(c["p2x"]) for c in madd], B),
                             ints_to_be([I(c["p2y"]) for c in madd], B)], axis=1)
        out, inf = ctx.mul_add2(curve, k1, p1, k2, p2)
        for i, c in enumerate(madd):
            assert _res_from(out, inf, i, B) == res_xy(c["r"]), ("muladd", curve, c)
            n_checked += 1
        # the P1 = G form (comb + ladder), on the cases whose first point is G
        from oracle import ec_oracle as O
        cur = O.get_curve(curve)
        if curve == "ed25519":
            gx, gy = cur.g.x, cur.g.y
        else:
            gx, gy = cur.g.x, cur.g.y
        gi = [i for i, c in enumerate(madd) if (I(c["p1x"]), I(c["p1y"])) == (gx, gy)]
        if gi:
            out, inf = ctx.mul_add2(curve, k1[gi], None, k2[gi], p2[gi])
            for j, i in enumerate(gi):
                assert _res_from(out, inf, j, B) == res_xy(madd[i]["r"]), ("muladd-G", curve, madd[i])
                n_checked += 1
    return n_checked


def check_x25519_golden(ctx):
    cases = mul_cases("curve25519")
    ks = ints_to_be([I(c["k"]) for c in cases], 32)
    xs = ints_to_be([I(c["px"]) for c in cases], 32)
    out, inf = ctx.x25519(ks, xs)
    for i, c in enumerate(cases):
        assert _res_from(out, inf, i, 32, xonly=True) == res_xy(c["r"]), c
    return len(cases)


def check_verify_golden(ctx, curve):
    """golden ECDSA verify tuples, grouped by (hash length, msgBitLength)."""
    B, NB = FIELD_BYTES[curve], ORDER_BYTES[curve]
    groups = {}
    for c in verify_cases(curve):
        key = (len(c["z"]) // 2, c.get("msgBitLength", 0))
        groups.setdefault(key, []).append(c)
    n_checked = 0
    for (hlen, mbits), cs in groups.items():
        hashes = ints_to_be([I(c["z"]) for c in cs], hlen)
        r = ints_to_be([I(c["r"]) for c in cs], NB)
        s = ints_to_be([I(c["s"]) for c in cs], NB)
        pub = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], B),
                              ints_to_be([I(c["qy"]) for c in cs], B)], axis=1)
        nlimb_bits = 32 * ((NB * 8 + 31) // 32)
        from oracle import ec_oracle as O
        nbits = O.get_curve(curve).n.bit_length()
        bits = mbits or hlen * 8
        if hlen * 8 - max(0, bits - nbits) > nlimb_bits:
            continue                      # outside the C ABI's documented domain
        ok = ctx.ecdsa_verify(curve, hashes, r, s, pub, msg_bits=mbits)
        for i, c in enumerate(cs):
            assert ok[i] == (1 if c["ok"] else 0), (curve, c, int(ok[i]))
            n_checked += 1
    return n_checked


def check_decompress_golden(ctx, curve):
    """pointFromX / pointFromY goldens (valid and invalid abscissae, both parities)"""
    from golden_util import load
    B = FIELD_BYTES[curve]
    cases = load("decompress_%s.json" % curve)
    v = ints_to_be([I(c["v"]) for c in cases], B)
    odd = np.array([1 if c["odd"] else 0 for c in cases], np.uint8)
    out, ok = ctx.decompress(curve, v, odd)
    n_inv = 0
    for i, c in enumerate(cases):
        if "invalid" in c["r"]:
            assert ok[i] == 0, (curve, c)
            n_inv += 1
        else:
            assert ok[i] == 1, (curve, c)
            got = (int.from_bytes(out[i, :B].tobytes(), "big"), int.from_bytes(out[i, B:].tobytes(), "big"))
            assert got == (I(c["r"]["x"]), I(c["r"]["y"])), (curve, c)
    assert n_inv > 3
    return len(cases)


def check_ed_from_x_golden(ctx):
    """EdwardsCurve#pointFromX goldens (tools/gen_golden.js genEdFromX) through ellgpu_decompress
    with bit 1 of the parity byte set"""
    from golden_util import load
    cases = load("fromx_ed25519.json")
    v = ints_to_be([I(c["v"]) % (1 << 256) for c in cases], 32)
    odd = np.array([3 if c["odd"] else 2 for c in cases], np.uint8)
    out, ok = ctx.decompress("ed25519", v, odd)
    n_inv = 0
    for i, c in enumerate(cases):
        if "invalid" in c["r"]:
            assert ok[i] == 0 and not out[i].any(), c
            n_inv += 1
        else:
            assert ok[i] == 1, c
            assert out[i].tobytes().hex() == c["r"]["x"] + c["r"]["y"], c
    assert n_inv > 10
    return len(cases)


def check_eddsa_golden(ctx):
    """EDDSA#verify goldens: the reference's sign.input vectors + corrupted / malformed ones"""
    from golden_util import load
    cases = load("eddsa_verify_ed25519.json")
    msgs = [bytes.fromhex(c["msg"]) for c in cases]
    sigs = np.frombuffer(b"".join(bytes.fromhex(c["sig"]) for c in cases), np.uint8).reshape(-1, 64)
    pubs = np.frombuffer(b"".join(bytes.fromhex(c["pub"]) for c in cases), np.uint8).reshape(-1, 32)
    ok, err = ctx.eddsa_verify(msgs, sigs, pubs)
    for i, c in enumerate(cases):
        if "throws" in c:
            assert err[i] == 1 and ok[i] == 0, c
        else:
            assert err[i] == 0 and bool(ok[i]) == c["ok"], c
    # uniform-length form on the subset that shares the most common length
    from collections import Counter
    L = Counter(len(m) for m in msgs).most_common(1)[0][0]
    idx = [i for i, m in enumerate(msgs) if len(m) == L]
    if L > 0 and len(idx) > 1:
        arr = np.frombuffer(b"".join(msgs[i] for i in idx), np.uint8).reshape(len(idx), L)
        ok2, err2 = ctx.eddsa_verify(arr, sigs[idx], pubs[idx])
        assert np.array_equal(ok2, ok[idx]) and np.array_equal(err2, err[idx])
    return len(cases)


def check_signdet_golden(ctx, curve):
    """EC#sign with the reference's own HmacDRBG nonces: (z, d, canonical) -> (r, s, recoveryParam)"""
    from golden_util import load
    NB = ORDER_BYTES[curve]
    groups = {}
    for c in load("signdet_%s.json" % curve):
        groups.setdefault((len(c["z"]) // 2, c["canonical"]), []).append(c)
    total = 0
    for (zlen, canonical), cs in sorted(groups.items()):
        z = np.frombuffer(b"".join(bytes.fromhex(c["z"]) for c in cs), np.uint8).reshape(-1, zlen)
        d = np.frombuffer(b"".join(bytes.fromhex(c["d"]) for c in cs), np.uint8).reshape(-1, NB)
        r, s, rec, ok = ctx.ecdsa_sign_det(curve, z, d, canonical=canonical)
        for i, c in enumerate(cs):
            assert ok[i] == 1, c["note"]
            assert (r[i].tobytes().hex(), s[i].tobytes().hex(), int(rec[i])) == (c["r"], c["s"], c["recid"]), c["note"]
            total += 1
    # a private key >= n signs like its residue (KeyPair#_importPrivate reduces it, ec/key.js:91-96):
    # the DRBG is seeded with d mod n
    n_int = int.from_bytes(bytes.fromhex(load("curves.json")[curve]["n"].rjust(2 * NB, "0")), "big")
    d0 = 0x1234567
    ds = [d0] + [d0 + m * n_int for m in (1, 100) if d0 + m * n_int < 1 << (8 * NB)]
    if len(ds) > 1:
        z = np.tile(np.frombuffer(bytes(range(1, 33)), np.uint8), (len(ds), 1))
        r, s, rec, ok = ctx.ecdsa_sign_det(curve, z, ints_to_be(ds, NB))
        assert ok.all() and (r == r[0]).all() and (s == s[0]).all() and (rec == rec[0]).all()
        total += len(ds) - 1
    return total


def check_recover_golden(ctx, curve):
    """EC#recoverPubKey goldens: points, infinity, and status 2 exactly where the reference
    throws ('invalid point' / 'Unable to find sencond key candinate')"""
    from golden_util import load
    B, NB = FIELD_BYTES[curve], ORDER_BYTES[curve]
    groups = {}
    for c in load("recover_%s.json" % curve):
        groups.setdefault(len(c["z"]) // 2, []).append(c)
    total = 0
    for zlen, cs in sorted(groups.items()):
        z = np.frombuffer(b"".join(bytes.fromhex(c["z"]) for c in cs), np.uint8).reshape(-1, zlen)
        r = np.frombuffer(b"".join(bytes.fromhex(c["r"]) for c in cs), np.uint8).reshape(-1, NB)
        s = np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in cs), np.uint8).reshape(-1, NB)
        j = np.array([c["j"] for c in cs], np.uint8)
        xy, st = ctx.ecdsa_recover(curve, z, r, s, j)
        for i, c in enumerate(cs):
            if "throws" in c:
                assert st[i] == 2 and not xy[i].any(), c
            elif c["q"].get("inf"):
                assert st[i] == 1, c
            else:
                assert st[i] == 0, c
                assert xy[i].tobytes().hex() == c["q"]["x"] + c["q"]["y"], c
            total += 1
    # outside the domain: r = 0 and r = n are reported, not computed
    n_int = int.from_bytes(bytes.fromhex(load("curves.json")[curve]["n"].rjust(2 * NB, "0")), "big")
    z = np.zeros((2, 32), np.uint8)
    r = ints_to_be([0, n_int], NB)
    s = ints_to_be([5, 5], NB)
    xy, st = ctx.ecdsa_recover(curve, z, r, s, np.array([0, 1], np.uint8))
    assert list(st) == [3, 3] and not xy.any()
    return total


DECODE_STATUS = {"Unknown point format": 1, "invalid point": 2, "Assertion failed": 3}
VALIDATE_STATUS = {None: 0, "Invalid public key": 1, "Public key is not a point": 2,
                   "Public key * N != O": 3}


def check_codec_golden(ctx, curve):
    """decodePoint / encode / KeyPair#validate goldens (tools/gen_golden.js genCodec): the status
    byte names the exception the reference throws, results are byte-identical"""
    from golden_util import load
    B = FIELD_BYTES[curve]
    g = load("codec_%s.json" % curve)
    total = 0
    groups = {}
    for c in g["decode"]:
        groups.setdefault(len(c["enc"]) // 2, []).append(c)
    for ln, cs in sorted(groups.items()):
        enc = np.frombuffer(b"".join(bytes.fromhex(c["enc"]) for c in cs), np.uint8).reshape(-1, ln)
        xy, st = ctx.decode_points(curve, enc)
        for i, c in enumerate(cs):
            if "throws" in c["r"]:
                want = 2 if curve == "ed25519" else DECODE_STATUS[c["r"]["throws"]]
                assert st[i] == want and not xy[i].any(), c
            else:
                assert st[i] == 0, c
                assert xy[i].tobytes().hex() == c["r"]["x"] + c["r"]["y"], c
            total += 1
    xy = np.frombuffer(b"".join(bytes.fromhex(c["x"] + c["y"]) for c in g["encode"]), np.uint8).reshape(-1, 2 * B)
    comp = ctx.encode_points(curve, xy, compact=True)
    for i, c in enumerate(g["encode"]):
        assert comp[i].tobytes().hex() == c["compact"], c
    if curve != "ed25519":
        full = ctx.encode_points(curve, xy, compact=False)
        for i, c in enumerate(g["encode"]):
            assert full[i].tobytes().hex() == c["full"], c
    total += len(g["encode"])
    # (a coordinate >= 2^(8B) -- keyFromPublic takes any integer -- has no fixed-width form)
    vcases = [c for c in g["validate"] if len(c["x"]) == 2 * B and len(c["y"]) == 2 * B]
    xy = np.frombuffer(b"".join(bytes.fromhex(c["x"] + c["y"]) for c in vcases), np.uint8).reshape(-1, 2 * B)
    st_full = ctx.validate(curve, xy, check_order=True)
    st_eq = ctx.validate(curve, xy, check_order=False)
    for i, c in enumerate(vcases):
        if curve == "ed25519":
            want_eq = 0 if c["on_curve"] else 2
            want_full = want_eq if (want_eq or c["order_ok"]) else 3
        else:
            want_full = VALIDATE_STATUS[c["reason"]]
            assert (want_full == 0) == c["result"]
            want_eq = want_full if want_full != 3 else 0
        assert st_eq[i] == want_eq and st_full[i] == want_full, (c, st_eq[i], st_full[i])
    total += len(vcases)
    # the caller's infinity flags come back as 'Invalid public key'
    inf = np.zeros(len(xy), np.uint8)
    inf[::3] = 1
    st = ctx.validate(curve, xy, inf=inf, check_order=False)
    assert (st[::3] == 1).all() and (st[1::3] == st_eq[1::3]).all()
    return total


def check_wire_golden(ctx, curve):
    """EC#verify on DER signatures + SEC1 keys, Signature#toDER / _importDER goldens
    (tools/gen_golden.js genWire): results and the exception the reference throws"""
    from golden_util import load
    NB = ORDER_BYTES[curve]
    g = load("wire_%s.json" % curve)
    total = 0
    groups = {}
    for c in g["verify"]:
        groups.setdefault((len(c["key"]) // 2, len(c["z"]) // 2), []).append(c)
    for (klen, zlen), cs in sorted(groups.items()):
        if klen == 0:
            continue
        z = np.frombuffer(b"".join(bytes.fromhex(c["z"]) for c in cs), np.uint8).reshape(-1, zlen)
        keys = np.frombuffer(b"".join(bytes.fromhex(c["key"]) for c in cs), np.uint8).reshape(-1, klen)
        ok, err = ctx.ecdsa_verify_wire(curve, z, [bytes.fromhex(c["der"]) for c in cs], keys)
        for i, c in enumerate(cs):
            if "throws" in c:
                want = 4 if c["throws"] == "Signature without r or s" else DECODE_STATUS[c["throws"]]
                assert err[i] == want and ok[i] == 0, (c, err[i], ok[i])
            else:
                assert err[i] == 0 and bool(ok[i]) == c["ok"], (c, err[i], ok[i])
            total += 1
    r = ints_to_be([int(c["r"], 16) for c in g["der"]], NB)
    s = ints_to_be([int(c["s"], 16) for c in g["der"]], NB)
    ders = ctx.sig_to_der(curve, r, s)
    for d, c in zip(ders, g["der"]):
        assert d.hex() == c["der"], c
    total += len(ders)
    pr, ps, st = ctx.sig_from_der(curve, [bytes.fromhex(c["der"]) for c in g["parse"]])
    for i, c in enumerate(g["parse"]):
        if c.get("bad"):
            assert st[i] == 1 and not pr[i].any() and not ps[i].any(), c
        else:
            ri, si = int(c["r"], 16), int(c["s"], 16)
            if max(ri.bit_length(), si.bit_length()) > 8 * NB:
                assert st[i] == 2, c
            else:
                assert st[i] == 0, c
                assert int.from_bytes(pr[i].tobytes(), "big") == ri and int.from_bytes(ps[i].tobytes(), "big") == si, c
    total += len(g["parse"])
    # round trip at scale: toDER then _importDER gives the same (r, s)
    # (r = 0 encodes as 02 01 00, which _importDER itself rejects as a leading zero)
    rr, ss, st = ctx.sig_from_der(curve, ders)
    nz = r.any(axis=1)
    assert (st[nz] == 0).all() and (st[~nz] == 1).all()
    assert np.array_equal(rr[nz], r[nz]) and np.array_equal(ss[nz], s[nz])
    return total


def check_der_fuzz(ctx):
    """Signature#_importDER on 2.4 k mutated encodings (reference verdicts, der_fuzz_secp256k1.json)"""
    from golden_util import load
    cases = load("der_fuzz_secp256k1.json")
    pr, ps, st = ctx.sig_from_der("secp256k1", [bytes.fromhex(c["der"]) for c in cases])
    kinds = [0, 0, 0]
    for i, c in enumerate(cases):
        if c.get("bad"):
            want = 1
        else:
            ri, si = int(c["r"], 16), int(c["s"], 16)
            want = 2 if max(ri.bit_length(), si.bit_length()) > 256 else 0
            if want == 0:
                assert int.from_bytes(pr[i].tobytes(), "big") == ri and int.from_bytes(ps[i].tobytes(), "big") == si, c
        assert st[i] == want, (c, st[i])
        kinds[want] += 1
    assert min(kinds) > 50
    return len(cases)


def check_add_golden(ctx, curve):
    """Point#add goldens: random pairs, doublings, inverses, infinity on either side, order-2 /
    small-order points, off-curve operands (short curves)"""
    from golden_util import load
    B = FIELD_BYTES[curve]
    cases = load("add_%s.json" % curve)

    def pack(key):
        xy = b"".join(bytes(2 * B) if c[key].get("inf") and "x" not in c[key] else bytes.fromhex(c[key]["x"] + c[key]["y"])
                      for c in cases)
        inf = np.array([1 if (c[key].get("inf") and "x" not in c[key]) else 0 for c in cases], np.uint8)
        return np.frombuffer(xy, np.uint8).reshape(-1, 2 * B), inf
    p, pinf = pack("p")
    q, qinf = pack("q")
    out, inf = ctx.point_add(curve, p, q, pinf, qinf)
    for i, c in enumerate(cases):
        if "x" in c["r"]:
            assert out[i].tobytes().hex() == c["r"]["x"] + c["r"]["y"], c
            assert bool(inf[i]) == bool(c["r"].get("inf", False)), c
        else:
            assert inf[i] == 1, c
    # without flags every operand is finite: same answers on the finite subset
    out2, inf2 = ctx.point_add(curve, p, q)
    fin = (pinf == 0) & (qinf == 0)
    assert np.array_equal(out2[fin], out[fin]) and np.array_equal(inf2[fin], inf[fin])
    return len(cases)


def check_codec_random(ctx, curve, n=3000, seed=5):
    """decodePoint / encode / validate / Point#add on seeded random inputs against the oracle's
    restatement: on-curve points (G multiples), random bytes (mostly off-curve / undecodable),
    equal and opposite operands"""
    from oracle import ec_oracle as O
    cur = O.get_curve(curve)
    B = FIELD_BYTES[curve]
    rng = np.random.default_rng(seed)
    ks = [int.from_bytes(rng.bytes(B), "big") % (cur.n - 1) + 1 for _ in range(n)]
    pts, _ = ctx.mul_fixed(curve, ints_to_be(ks, B))
    # compressed encodings of valid points and of random x: decode, then re-encode
    enc = ctx.encode_points(curve, pts, compact=True)
    junk = np.frombuffer(rng.bytes(n * (1 + B)), np.uint8).reshape(n, 1 + B).copy()
    junk[:, 0] = 2 + (junk[:, 0] & 1)
    if curve == "p521":
        junk[:, 1] &= 1                                       # keep x below 2^521 half of the time
    both = np.concatenate([enc, junk])
    xy, st = ctx.decode_points(curve, both)
    assert (st[:n] == 0).all() and np.array_equal(xy[:n], pts)
    for i in range(n, 2 * n, 7):                              # the oracle on a sample of the random ones
        try:
            p = O.decode_point(cur, both[i].tobytes())
            assert st[i] == 0 and xy[i].tobytes() == p.x.to_bytes(B, "big") + p.y.to_bytes(B, "big"), i
        except ValueError:
            assert st[i] == 2 and not xy[i].any(), i
    ok = st == 0
    assert 0.3 * n < ok[n:].sum() < 0.7 * n                   # about half of all x lift
    assert np.array_equal(ctx.encode_points(curve, xy[ok], compact=True), both[ok])
    # validate: decoded points are on the curve, perturbed ones are not
    assert not ctx.validate(curve, xy[ok], check_order=False).any()
    bad = xy[ok].copy()
    bad[:, 2 * B - 1] ^= 1
    assert (ctx.validate(curve, bad, check_order=False) == 2).all()
    # Point#add: P + Q, P + P, P + (-P), and off-curve operands, against the oracle
    q = np.roll(pts, 1, axis=0)
    q[::5] = pts[::5]                                         # P + P
    neg = (cur.p - np.array(be_to_ints(pts[1::5, B:]), dtype=object)) % cur.p
    q[1::5, :B] = pts[1::5, :B]
    q[1::5, B:] = ints_to_be([int(v) for v in neg], B)        # P + (-P)
    q[2::5] = np.frombuffer(rng.bytes(len(q[2::5]) * 2 * B), np.uint8).reshape(-1, 2 * B)
    if curve == "p521":
        q[2::5, 0] &= 1
        q[2::5, B] &= 1
    out, inf = ctx.point_add(curve, pts, q)
    px, py = be_to_ints(pts[:, :B]), be_to_ints(pts[:, B:])
    qx, qy = be_to_ints(q[:, :B]), be_to_ints(q[:, B:])
    for i in range(0, n, 3):
        r = cur.point(px[i], py[i]).add(cur.point(qx[i] % cur.p, qy[i] % cur.p))
        if r.inf:
            assert inf[i] == 1, i
        else:
            assert inf[i] == 0 and out[i].tobytes() == r.x.to_bytes(B, "big") + r.y.to_bytes(B, "big"), i
    assert (inf[1::5] == 1).all()
    return n


def check_ecdh(ctx, curve, n=24):
    """KeyPair#derive: a*(b*G) == b*(a*G), equal to the oracle's pub.mul(priv).getX(); an
    off-curve public point is refused as the reference's assert does; priv = n gives infinity"""
    from oracle import ec_oracle as O
    cur = O.get_curve(curve)
    B = FIELD_BYTES[curve]
    rng = np.random.default_rng(11)
    a = [int.from_bytes(rng.bytes(B), "big") % (cur.n - 1) + 1 for _ in range(n)]
    b = [int.from_bytes(rng.bytes(B), "big") % (cur.n - 1) + 1 for _ in range(n)]
    A, _ = ctx.mul_fixed(curve, ints_to_be(a, B))
    Bp, _ = ctx.mul_fixed(curve, ints_to_be(b, B))
    s1, st1 = ctx.ecdh_derive(curve, ints_to_be(a, B), Bp)
    s2, st2 = ctx.ecdh_derive(curve, ints_to_be(b, B), A)
    assert not st1.any() and not st2.any() and np.array_equal(s1, s2)
    for i in range(0, n, 5):
        pub = cur.point(int.from_bytes(Bp[i, :B].tobytes(), "big"), int.from_bytes(Bp[i, B:].tobytes(), "big"))
        assert cur.validate(pub)
        assert pub.mul(a[i]).x == int.from_bytes(s1[i].tobytes(), "big")
    bad = Bp.copy()
    bad[::3, 2 * B - 1] ^= 1
    ks = ints_to_be([cur.n if i % 4 == 1 else a[i] for i in range(n)], B)
    _, st = ctx.ecdh_derive(curve, ks, bad)
    for i in range(n):
        assert st[i] == (1 if i % 3 == 0 else (2 if i % 4 == 1 else 0)), i
    return n


def check_eddsa_sign_golden(ctx):
    """EDDSA#sign / keyFromSecret goldens: sign.input vectors + seeded block-boundary lengths;
    the signatures must also verify"""
    from golden_util import load
    cases = load("eddsa_sign_ed25519.json")
    msgs = [bytes.fromhex(c["msg"]) for c in cases]
    sec = np.frombuffer(b"".join(bytes.fromhex(c["secret"]) for c in cases), np.uint8).reshape(-1, 32)
    sig, pub = ctx.eddsa_sign(msgs, sec)
    for i, c in enumerate(cases):
        assert sig[i].tobytes().hex() == c["sig"], c["note"]
        assert pub[i].tobytes().hex() == c["pub"], c["note"]
    ok, err = ctx.eddsa_verify(msgs, sig, pub)
    assert ok.all() and not err.any()
    # uniform-length form
    from collections import Counter
    L = Counter(len(m) for m in msgs).most_common(1)[0][0]
    idx = [i for i, m in enumerate(msgs) if len(m) == L]
    if len(idx) > 1:
        arr = np.frombuffer(b"".join(msgs[i] for i in idx), np.uint8).reshape(len(idx), L)
        sig2, pub2 = ctx.eddsa_sign(arr, sec[idx])
        assert np.array_equal(sig2, sig[idx]) and np.array_equal(pub2, pub[idx])
    return len(cases)


def check_sign_golden(ctx, curve):
    """EC#sign with supplied nonces: (hash, d, k) -> (r, s, recoveryParam) or 'next nonce'"""
    from golden_util import load
    NB = ORDER_BYTES[curve]
    groups = {}
    for c in load("sign_%s.json" % curve):
        groups.setdefault((len(c["z"]) // 2, c["canonical"]), []).append(c)
    n_checked = 0
    for (hl, canon), cs in groups.items():
        r, s, rec, ok = ctx.ecdsa_sign(curve, ints_to_be([I(c["z"]) for c in cs], hl),
                                       ints_to_be([I(c["d"]) for c in cs], NB),
                                       ints_to_be([I(c["k"]) for c in cs], NB), canonical=canon)
        for i, c in enumerate(cs):
            if c.get("rejected"):
                assert ok[i] == 0, (curve, c)
            else:
                got = (int.from_bytes(r[i].tobytes(), "big"), int.from_bytes(s[i].tobytes(), "big"), int(rec[i]))
                assert ok[i] == 1 and got == (I(c["r"]), I(c["s"]), c["recid"]), (curve, c)
            n_checked += 1
    return n_checked


def custom_curves():
    """tests/golden/custom_short.json (tools/gen_golden_custom.js): user-defined short curves"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "custom_short.json")) as f:
        return json.load(f)


def check_custom_short_golden(ctx, spec):
    """Point#mul, mulAdd / jmulAdd, Point#add and JPoint#dblp of the reference on a user-defined
    short curve (run-time prime, arbitrary a) against ellgpu_curve_define_short's curve id."""
    p, a, b = I(spec["p"]), I(spec["a"]), I(spec["b"])
    cid = ctx.define_short(p, a, b)
    assert cid >= 16 and ctx.define_short(p, a, b) == cid            # same parameters, same id

    def xy(list_of_pts):
        return np.concatenate([ints_to_be([I(q["x"]) for q in list_of_pts], 32),
                               ints_to_be([I(q["y"]) for q in list_of_pts], 32)], axis=1)

    def want(r):
        return None if r.get("inf") else (I(r["x"]), I(r["y"]))
    cases = spec["cases"]
    mul = [c for c in cases if c["op"] == "mul"]
    out, inf = ctx.mul_var(cid, ints_to_be([I(c["k"]) for c in mul], 32), xy([c["p"] for c in mul]))
    for i, c in enumerate(mul):
        assert _res_from(out, inf, i, 32) == want(c["r"]), ("mul", spec["name"], c)
    madd = [c for c in cases if c["op"] == "muladd"]
    out, inf = ctx.mul_add2(cid, ints_to_be([I(c["k1"]) for c in madd], 32), xy([c["p1"] for c in madd]),
                            ints_to_be([I(c["k2"]) for c in madd], 32), xy([c["p2"] for c in madd]))
    for i, c in enumerate(madd):
        assert c["r"] == c["rj"]
        assert _res_from(out, inf, i, 32) == want(c["r"]), ("muladd", spec["name"], c)
    add = [c for c in cases if c["op"] == "add"]
    zero = {"x": "0", "y": "0"}
    out, inf = ctx.point_add(cid, xy([zero if c["p1"].get("inf") else c["p1"] for c in add]),
                             xy([zero if c["p2"].get("inf") else c["p2"] for c in add]),
                             inf1=np.array([1 if c["p1"].get("inf") else 0 for c in add], np.uint8),
                             inf2=np.array([1 if c["p2"].get("inf") else 0 for c in add], np.uint8))
    for i, c in enumerate(add):
        assert _res_from(out, inf, i, 32) == want(c["r"]), ("add", spec["name"], c)
    # dblp(pow) = 2^pow * P
    dbl = [c for c in cases if c["op"] == "dblp"]
    out, inf = ctx.mul_var(cid, ints_to_be([1 << c["pow"] for c in dbl], 32), xy([c["p"] for c in dbl]))
    for i, c in enumerate(dbl):
        assert _res_from(out, inf, i, 32) == want(c["r"]), ("dblp", spec["name"], c)
    # operands off the curve are reported, not guessed: every multiplicand with y + 1 (never on the
    # curve together with (x, y) unless y = (p - 1) / 2) -> out_inf = 2, result zeroed
    off = xy([c["p"] for c in mul])
    really_off = [(I(c["p"]["y"]) + 1) % p != (-I(c["p"]["y"])) % p for c in mul]
    off[:, 32:] = ints_to_be([(I(c["p"]["y"]) + 1) % p for c in mul], 32)
    out, inf = ctx.mul_var(cid, ints_to_be([I(c["k"]) for c in mul], 32), off)
    for i, c in enumerate(mul):
        if really_off[i]:
            assert inf[i] == 2 and not out[i].any(), ("off-curve mul", spec["name"], c)
    out, inf = ctx.mul_add2(cid, ints_to_be([I(c["k"]) for c in mul], 32), xy([c["p"] for c in mul]),
                            ints_to_be([I(c["k"]) for c in mul], 32), off)
    assert all(inf[i] == 2 and not out[i].any() for i in range(len(mul)) if really_off[i]), spec["name"]
    return len(mul) + len(madd) + len(add) + len(dbl)


def custom_edwards_curves():
    """tests/golden/custom_edwards.json (tools/gen_golden_custom.js): user-defined Edwards curves"""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "custom_edwards.json")) as f:
        return json.load(f)


def check_custom_edwards_golden(ctx, spec):
    """Point#mul, Point#add and dbl of the reference on a user-defined (twisted) Edwards curve --
    the projective _projDbl / _projAdd for a != -1, the extended forms for a = -1 -- against
    ellgpu_curve_define_edwards's curve id; k1*P1 + k2*P2 against the sum of the two products"""
    p, a, d = I(spec["p"]), I(spec["a"]), I(spec["d"])
    cid = ctx.define_edwards(p, a, d)
    assert cid >= 16 and ctx.define_edwards(p, a, d) == cid

    def xy(pts):
        return np.concatenate([ints_to_be([I(q["x"]) for q in pts], 32),
                               ints_to_be([I(q["y"]) for q in pts], 32)], axis=1)
    cases = spec["cases"]
    mul = [c for c in cases if c["op"] == "mul"]
    ks = ints_to_be([I(c["k"]) for c in mul], 32)
    out, inf = ctx.mul_var(cid, ks, xy([c["p"] for c in mul]))
    assert not inf.any()
    for i, c in enumerate(mul):
        assert _res_from(out, inf, i, 32) == (I(c["r"]["x"]), I(c["r"]["y"])), ("mul", spec["name"], c)
    add = [c for c in cases if c["op"] == "add"]
    out, inf = ctx.point_add(cid, xy([c["p1"] for c in add]), xy([c["p2"] for c in add]))
    for i, c in enumerate(add):
        assert _res_from(out, inf, i, 32) == (I(c["r"]["x"]), I(c["r"]["y"])), ("add", spec["name"], c)
    # the identity through the inf flags: (0, 1) + P = P
    flags = np.ones(len(add), np.uint8)
    out, inf = ctx.point_add(cid, xy([c["p1"] for c in add]), xy([c["p2"] for c in add]), inf1=flags)
    assert np.array_equal(out, xy([c["p2"] for c in add]))
    dbl = [c for c in cases if c["op"] == "dbl"]
    out, inf = ctx.point_add(cid, xy([c["p"] for c in dbl]), xy([c["p"] for c in dbl]))
    for i, c in enumerate(dbl):
        assert _res_from(out, inf, i, 32) == (I(c["r"]["x"]), I(c["r"]["y"])), ("dbl", spec["name"], c)
    # k1*P1 + k2*P2 == (k1*P1) + (k2*P2)
    m = len(mul) // 2
    got, _ = ctx.mul_add2(cid, ks[:m], xy([c["p"] for c in mul[:m]]), ks[m:2 * m], xy([c["p"] for c in mul[m:2 * m]]))
    want, _ = ctx.point_add(cid, xy([c["r"] for c in mul[:m]]), xy([c["r"] for c in mul[m:2 * m]]))
    assert np.array_equal(got, want)
    # operands off the curve (x + 1 instead of x) are reported, not guessed
    off = xy([c["p"] for c in mul])
    off[:, :32] = ints_to_be([(I(c["p"]["x"]) + 1) % p for c in mul], 32)
    offc = [(a * ((I(c["p"]["x"]) + 1) ** 2) + I(c["p"]["y"]) ** 2 - 1 - d * ((I(c["p"]["x"]) + 1) ** 2) * I(c["p"]["y"]) ** 2) % p != 0
            for c in mul]
    out, inf = ctx.mul_var(cid, ks, off)
    assert any(offc)
    for i in range(len(mul)):
        assert (inf[i] == 2 and not out[i].any()) if offc[i] else inf[i] == 0, ("off-curve mul", spec["name"], i)
    out, inf = ctx.mul_add2(cid, ks, xy([c["p"] for c in mul]), ks, off)
    assert all(inf[i] == 2 for i in range(len(mul)) if offc[i])
    return len(mul) + len(add) + len(dbl) + m


def comb_boundary_scalars(bits, cb, count, seed=22):
    """scalars whose cb-bit windows sit on the signed comb's recoding boundaries: 0, 1, 2^(cb-1) - 1,
    2^(cb-1) (the largest positive digit), 2^(cb-1) + 1 (the first one that is taken negative and
    carries), 2^cb - 2, 2^cb - 1 (with a carry arriving: digit 0 and a carry out) -- mixed so that
    carries run through several windows; plus the all-ones and the alternating patterns"""
    import random
    rnd = random.Random(seed)
    half, mask = 1 << (cb - 1), (1 << cb) - 1
    vals = [0, 1, half - 1, half, half + 1, mask - 1, mask]
    nw = (bits + cb - 1) // cb
    out = [0, 1, (1 << bits) - 1, int("5" * (bits // 4), 16), int("a" * (bits // 4), 16)]
    for v in vals:
        out.append(sum(v << (cb * w) for w in range(nw)) & ((1 << bits) - 1))
    while len(out) < count:
        k = 0
        for w in range(nw):
            v = rnd.choice(vals) if rnd.random() < 0.8 else rnd.getrandbits(cb)
            k |= v << (cb * w)
        out.append(k & ((1 << bits) - 1))
    return out


def check_offcurve_golden(ctx, curve):
    """offcurve_<curve>.json: operands that are NOT on the curve.  The reference computes with
    them (the file holds what it answers); the engine's C ABI must report every such item as
    outside its domain -- out_inf = 2 with a zeroed result, out_ok = 2 -- never a guess, and
    must go on computing the on-curve items of the same batch.  Returns the number of items."""
    from golden_util import load
    B = FIELD_BYTES[curve]
    cases = load("offcurve_%s.json" % curve)

    def pts(cs, x, y):
        return np.concatenate([ints_to_be([I(c[x]) for c in cs], B), ints_to_be([I(c[y]) for c in cs], B)], axis=1)

    def point_results(out, inf, cs, what):
        for i, c in enumerate(cs):
            if c["on"]:
                assert inf[i] in (0, 1) and _res_from(out, inf, i, B) == res_xy(c["r"]), (what, curve, c)
            else:
                assert inf[i] == 2 and not out[i].any(), (what, curve, c, int(inf[i]))

    total = 0
    var = [c for c in cases if c["op"] == "var"]
    out, inf = ctx.mul_var(curve, ints_to_be([I(c["k"]) for c in var], B), pts(var, "px", "py"))
    point_results(out, inf, var, "var")
    assert sum(1 for c in var if c["on"]) >= 2 and sum(1 for c in var if not c["on"]) >= 10
    total += len(var)
    madd = [c for c in cases if c["op"] == "muladd"]
    k1 = ints_to_be([I(c["k1"]) for c in madd], B)
    k2 = ints_to_be([I(c["k2"]) for c in madd], B)
    out, inf = ctx.mul_add2(curve, k1, pts(madd, "p1x", "p1y"), k2, pts(madd, "p2x", "p2y"))
    point_results(out, inf, madd, "muladd")
    gi = [i for i, c in enumerate(madd) if c["g1"]]
    out, inf = ctx.mul_add2(curve, k1[gi], None, k2[gi], pts([madd[i] for i in gi], "p2x", "p2y"))
    point_results(out, inf, [madd[i] for i in gi], "muladd-G")
    total += len(madd) + len(gi)
    add = [c for c in cases if c["op"] == "add"]
    if add:                       # Point#add is one formula: equal off the curve too
        p = np.frombuffer(b"".join(bytes.fromhex(c["p"]["x"] + c["p"]["y"]) for c in add), np.uint8).reshape(-1, 2 * B)
        q = np.frombuffer(b"".join(bytes.fromhex(c["q"]["x"] + c["q"]["y"]) for c in add), np.uint8).reshape(-1, 2 * B)
        out, inf = ctx.point_add(curve, p, q)
        for i, c in enumerate(add):
            assert out[i].tobytes().hex() == c["r"]["x"] + c["r"]["y"], ("add", c)
        total += len(add)
    ver = [c for c in cases if c["op"] == "verify"]
    if not ver:
        return total
    NB = ORDER_BYTES[curve]
    hl = len(ver[0]["z"]) // 2
    from oracle import ec_oracle as O
    n = O.get_curve(curve).n
    z = ints_to_be([I(c["z"]) for c in ver], hl)
    r = ints_to_be([I(c["r"]) for c in ver], NB)
    s = ints_to_be([I(c["s"]) for c in ver], NB)
    pub = pts(ver, "qx", "qy")
    ok, st = ctx.ecdsa_verify(curve, z, r, s, pub, status=True)
    assert np.array_equal(ok, ctx.ecdsa_verify(curve, z, r, s, pub))       # the status is optional
    assert set(np.unique(ok).tolist()) <= {0, 1}                           # a mask, whatever the keys are
    n_dom = 0
    for i, c in enumerate(ver):
        in_range = 0 < I(c["r"]) < n and 0 < I(c["s"]) < n
        if c["on"] or not in_range:
            # on the curve, or rejected before the key is touched (ec/index.js:199-202): a verdict
            assert (ok[i], st[i]) == (1 if c["ok"] else 0, 0), (curve, c, int(ok[i]), int(st[i]))
        else:
            # outside the engine's domain: verdict 0 (never "accept"), status 2
            assert (ok[i], st[i]) == (0, 2), (curve, c, int(ok[i]), int(st[i]))
            n_dom += 1
    assert n_dom >= 9 and sum(1 for c in ver if c["ok"] and not c["on"]) >= 8
    # the same tuples through the wire form: DER signatures + uncompressed keys
    inr = [i for i, c in enumerate(ver) if 0 < I(c["r"]) < n and 0 < I(c["s"]) < n]
    ders = ctx.sig_to_der(curve, r[inr], s[inr])
    keys = np.concatenate([np.full((len(inr), 1), 4, np.uint8), pub[inr]], axis=1)
    wok, werr = ctx.ecdsa_verify_wire(curve, z[inr], ders, keys)
    for j, i in enumerate(inr):
        c = ver[i]
        if c["on"]:
            assert (wok[j], werr[j]) == (1 if c["ok"] else 0, 0), ("wire", curve, c)
        else:
            assert (wok[j], werr[j]) == (0, 5), ("wire", curve, c, int(wok[j]), int(werr[j]))
    return total + len(ver) + len(inr)


def check_exceptional_keys(ctx, curve, seed=9):
    """Directed ECDSA tuples whose double-scalar multiplication walks through the group law's
    exceptional cases: keys that are multiples of G by 1, -1, 2, (for secp256k1) lambda, -lambda,
    lambda^2, (n - 1) / 2, with (u1, u2) chosen so that u1 G + u2 Q is O, G, -G, 2G, Q, or that a GLV
    half of u2 is 0 / 1 / a lattice vector, next to random ones.  Every tuple is built to VERIFY
    (r = x(R) mod n, s = r / u2, z = u1 s) unless R = O; the engine's verdicts and its k1 G + k2 Q
    results must equal the C port of the reference's algorithm item by item."""
    import random
    from oracle import c_oracle, ec_oracle as O
    cur = O.get_curve(curve)
    n, B = cur.n, FIELD_BYTES[curve]
    NB = ORDER_BYTES[curve]
    rnd = random.Random(seed)
    ds = [1, n - 1, 2, (n - 1) // 2, 3, rnd.randrange(1, n)]
    if cur.endo is not None:
        lam = cur.endo["lambda"]
        ds += [lam, n - lam, lam * lam % n, (lam + 1) % n, (lam - 1) % n]
    tuples = []                                     # (d, u1, u2)
    for d in ds:
        dinv = pow(d, -1, n)
        u2s = [1, 2, 3, n - 1, n - 2, rnd.randrange(1, n), rnd.randrange(1, 1 << 128), (1 << 128), (1 << 129) - 1]
        if cur.endo is not None:
            u2s += [lam, n - lam, (lam + 1) % n, lam * 2 % n]
        for u2 in u2s:
            for tgt in (0, 1, n - 1, 2, d % n, (n - d) % n, None):      # u1 + u2 d = tgt (None: random)
                u1 = rnd.randrange(n) if tgt is None else (tgt - u2 * d) % n
                tuples.append((d, u1, u2))
    m = len(tuples)
    dk = ints_to_be([t[0] for t in tuples], B)
    keys, kinf = c_oracle.mul(curve, dk)            # Q = d G (the oracle's fixed-base path)
    assert not kinf.any()
    k1 = ints_to_be([t[1] for t in tuples], B)
    k2 = ints_to_be([t[2] for t in tuples], B)
    want_xy, want_inf = c_oracle.mul_add(curve, k1, None, k2, keys)
    got_xy, got_inf = ctx.mul_add2(curve, k1, None, k2, keys)
    assert np.array_equal(got_inf, want_inf) and np.array_equal(got_xy, want_xy), (curve, "k1 G + k2 Q on exceptional keys")
    # the same (k1, k2, Q) with G passed as an ordinary point: the two-table ladder
    g = np.tile(np.concatenate([ints_to_be([cur.g.x], B), ints_to_be([cur.g.y], B)], axis=1), (m, 1))
    got2_xy, got2_inf = ctx.mul_add2(curve, k1, g, k2, keys)
    assert np.array_equal(got2_inf, want_inf) and np.array_equal(got2_xy, want_xy), (curve, "k1 P1 + k2 P2 on exceptional keys")
    # ECDSA tuples from them
    zs, rs, ss, expect = [], [], [], []
    for i, (d, u1, u2) in enumerate(tuples):
        if want_inf[i]:
            r = rnd.randrange(1, n)
            ok = False
        else:
            r = int.from_bytes(want_xy[i, :B].tobytes(), "big") % n
            ok = r != 0
            if r == 0:
                r = 1
        s = r * pow(u2, -1, n) % n
        z = u1 * s % n
        if s == 0:
            s, ok = 1, False
        zs.append(z); rs.append(r); ss.append(s); expect.append(ok)
    hl = NB if cur.n.bit_length() % 8 == 0 else NB - 1
    keep = [i for i in range(m) if zs[i].bit_length() <= 8 * hl]      # digests the reference would not shift
    z = ints_to_be([zs[i] for i in keep], hl)
    r = ints_to_be([rs[i] for i in keep], NB)
    s = ints_to_be([ss[i] for i in keep], NB)
    pub = keys[keep]
    want = c_oracle.verify(curve, z, r, s, pub)
    got = ctx.ecdsa_verify(curve, z, r, s, pub)
    assert np.array_equal(np.asarray(got), np.asarray(want)), (curve, "verify on exceptional keys")
    # s == 1 substitutions aside, the construction's own verdicts hold as well
    exp = np.array([1 if expect[i] else 0 for i in keep], np.uint8)
    assert int((np.asarray(want) == exp).sum()) >= len(keep) - 4
    assert int(exp.sum()) > len(keep) // 2 and int(np.asarray(want_inf).sum()) >= len(ds)
    return 2 * m + len(keep)


def check_deferred_calls(ctx):
    """ellgpu_ctx_defer / ellgpu_ctx_collect: a deferred few-item call gives the same bytes as the
    plain call -- after collect(), or after ANY other entry point on the context (which completes
    it first); collect() without a pending call, and defer() in front of a call too large for the
    pinned buffer, are harmless."""
    import random
    from oracle import ec_oracle as O
    rnd = random.Random(606)
    cur = O.get_curve("secp256k1")
    n = 5
    ks = ints_to_be([rnd.randrange(1, 1 << 256) for _ in range(n)], 32)
    pts, _ = ctx.mul_fixed("secp256k1", ints_to_be([rnd.randrange(1, cur.n) for _ in range(n)], 32))
    want, winf = ctx.mul_var("secp256k1", ks, pts)
    assert want.any()
    # 1. defer -> call -> collect
    ctx.defer()
    got, ginf = ctx.mul_var("secp256k1", ks, pts)
    ctx.collect()
    assert np.array_equal(got, want) and np.array_equal(ginf, winf)
    # 2. another entry point in between completes the pending call before it runs
    ctx.defer()
    got2, ginf2 = ctx.mul_var("secp256k1", ks, pts)
    other, _ = ctx.mul_fixed("secp256k1", ks)               # would overwrite the pinned buffer
    assert np.array_equal(got2, want) and np.array_equal(ginf2, winf)
    ctx.collect()                                           # nothing left to do
    assert np.array_equal(got2, want)
    fx, _ = ctx.mul_fixed("secp256k1", ks)
    assert np.array_equal(fx, other)
    # 3. collect() with nothing pending; defer() is consumed by the next call only
    ctx.collect()
    ctx.defer()
    ctx.collect()                                           # disarms
    got3, _ = ctx.mul_var("secp256k1", ks, pts)             # a plain call again
    assert np.array_equal(got3, want)
    # 4. a verify and an EdDSA call in the split form
    zs = ints_to_be([rnd.randrange(1 << 256) for _ in range(n)], 32)
    rs = ints_to_be([rnd.randrange(1, cur.n) for _ in range(n)], 32)
    ok0 = ctx.ecdsa_verify("secp256k1", zs, rs, rs, pts)
    ctx.defer()
    ok1 = ctx.ecdsa_verify("secp256k1", zs, rs, rs, pts)
    ctx.collect()
    assert np.array_equal(ok0, ok1)
    secrets = np.frombuffer(bytes(rnd.randrange(256) for _ in range(32 * 3)), np.uint8).reshape(3, 32)
    msgs = [b"", b"abc", bytes(range(200))]
    sig0, pub0 = ctx.eddsa_sign(msgs, secrets)
    ctx.defer()
    sig1, pub1 = ctx.eddsa_sign(msgs, secrets)
    ctx.collect()
    assert np.array_equal(sig0, sig1) and np.array_equal(pub0, pub1)
    ok2, err2 = ctx.eddsa_verify(msgs, sig0, pub0)
    ctx.defer()
    ok3, err3 = ctx.eddsa_verify(msgs, sig0, pub0)
    ctx.collect()
    assert ok2.all() and np.array_equal(ok2, ok3) and np.array_equal(err2, err3)
    # 4b. only the ARMING thread's next call is deferred: another thread's call that slips in between
    #     (the N-API worker running a Promise-form batch beside the JS thread) runs to completion
    import threading
    ctx.defer()
    box = []
    th = threading.Thread(target=lambda: box.append(ctx.mul_var("secp256k1", ks, pts)))
    th.start()
    th.join()
    assert np.array_equal(box[0][0], want) and np.array_equal(box[0][1], winf)      # complete when it returned
    got4, ginf4 = ctx.mul_var("secp256k1", ks, pts)                                  # this thread's: still deferred
    ctx.collect()
    assert np.array_equal(got4, want) and np.array_equal(ginf4, winf)
    # 5. a batch that does not fit the pinned buffer is simply not deferred
    m = 6000
    kk = np.tile(ks, (m // n, 1))
    pp = np.tile(pts, (m // n, 1))
    ctx.defer()
    big, _ = ctx.mul_var("secp256k1", kk, pp)
    assert np.array_equal(big[:n], want)                    # complete already
    ctx.collect()
    assert np.array_equal(big[-n:], want)


def check_x25519_derive(ctx, count=96):
    """ellgpu_x25519_derive = KeyPair#derive on curve25519 (ec/key.js:102-107): status 1 exactly where
    x^3 + 486662 x^2 + x is a non-residue mod 2^255 - 19 (the reference's validate throws out of its
    square root there), else the ladder's x -- the same bytes as ellgpu_x25519_ladder -- and status 2
    where that call reports infinity.  Small and large batches (the row layer and the one-lane kernels)."""
    import random
    rnd = random.Random(2519)
    p = 2 ** 255 - 19
    xs = [0, 1, 2, 3, 4, 9, p - 1, p - 2, p + 5, (1 << 256) - 1] + [rnd.randrange(p) for _ in range(count)]
    ks = [rnd.randrange(1 << 255) for _ in xs]
    ks[3] = 0
    ks[4] = 1
    kb, xb = ints_to_be(ks, 32), ints_to_be(xs, 32)
    want_bad = []
    for x in xs:
        x %= p
        rhs = (x * x * x + 486662 * x * x + x) % p
        want_bad.append(0 if rhs == 0 or pow(rhs, (p - 1) // 2, p) == 1 else 1)
    assert 20 < sum(want_bad) < len(xs) - 20
    ref_x, ref_inf = ctx.x25519(kb, xb)
    for reps in (1, 40):                                   # 106 items; 4 240: above the row layer's batch sizes
        kk, xx = np.tile(kb, (reps, 1)), np.tile(xb, (reps, 1))
        out, st = ctx.x25519_derive(kk, xx)
        for i in range(len(xs) * reps):
            j = i % len(xs)
            assert st[i] == (1 if want_bad[j] else (2 if ref_inf[j] else 0)), (reps, i, hex(xs[j]))
            assert np.array_equal(out[i], ref_x[j]), (reps, i)
    # one item at a time (install()'s call)
    for j in range(12):
        out, st = ctx.x25519_derive(kb[j:j + 1], xb[j:j + 1])
        assert st[0] == (1 if want_bad[j] else (2 if ref_inf[j] else 0)) and np.array_equal(out[0], ref_x[j]), j
    return len(xs)
