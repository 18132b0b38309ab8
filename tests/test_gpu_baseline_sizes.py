"""Parity at BASELINE.json's stated sizes, inside the driver's `-m gpu` suite.

Every config runs at its full batch through the C ABI with >= 10 240 results compared item by
item with the oracle (oracle/ec_oracle.c -- itself pinned to the reference's golden vectors by
tests/test_c_oracle.py -- on the host threads; SURVEY.md 8d "Parity sampling"), a stride of the
whole batch included so that every region of the launch is sampled, and size-independent
properties over ALL items where the domain offers one.  Bounded: the oracle legs are ~1-3 s each
on the GPU box's 16 usable threads."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import elliptic_amd  # noqa: E402
from oracle import c_oracle as C  # noqa: E402

pytestmark = pytest.mark.gpu
SAMPLE = 10240


@pytest.fixture(scope="module")
def ctx():
    c = elliptic_amd.Context(0)
    yield c
    c.close()


def rnd(seed, n, w):
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(n * w), dtype=np.uint8).reshape(n, w).copy()


def sample_idx(n, m=SAMPLE):
    """first m/2 items + a stride over the whole batch + the last items"""
    head = np.arange(min(m // 2, n))
    stride = np.arange(0, n, max(1, n // (m // 2 - 64)))[: m // 2 - 64]
    tail = np.arange(max(0, n - 64), n)
    return np.unique(np.concatenate([head, stride, tail]))


def threads():
    return C._usable_cpus()


def test_config2_secp256k1_fixed_base_1M(ctx):
    n = 1 << 20
    k = rnd("bl:cfg2", n, 32)
    k[:8] = 0
    k[1, 31] = 1                                                  # edge scalars: 0, 1, n-1, n, n+1, 2^256-1 ...
    N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    for j, v in enumerate((N - 1, N, N + 1, (1 << 256) - 1, 2)):
        k[2 + j] = np.frombuffer(v.to_bytes(32, "big"), np.uint8)
    xy, inf = ctx.mul_fixed("secp256k1", k)
    idx = sample_idx(n)
    want, winf = C.mul_mt("secp256k1", k[idx], None, threads())
    assert np.array_equal(inf[idx], winf) and np.array_equal(xy[idx], want)
    assert inf[0] == 1 and inf[3] == 1 and inf[:8].sum() == 3     # 0*G, n*G and the other zero row


@pytest.mark.parametrize("curve", ["secp256k1", "p256"])
def test_signed_comb_window_boundaries(ctx, curve):
    """the signed 22-bit comb of the 256-bit curves: 4 096 scalars whose windows sit on the
    recoding's boundaries (largest positive digit, first negative one, carries through several
    windows, a carry into the top window) -- G*k against the C oracle, item by item, and the
    same scalars through k1*G + k2*P (the comb onto the ladder's accumulator)"""
    import parity_checks as PC
    from elliptic_amd import ints_to_be
    ks = PC.comb_boundary_scalars(256, 22, 4096)
    k = ints_to_be(ks, 32)
    xy, inf = ctx.mul_fixed(curve, k)
    want, winf = C.mul_mt(curve, k, None, threads())
    assert np.array_equal(inf, winf) and np.array_equal(xy, want)
    # k*G + 1*G == (k + 1)*G through mul_add2 with P1 = G (NULL)
    one = ints_to_be([1] * len(ks), 32)
    g, _ = ctx.mul_fixed(curve, one)
    s_xy, s_inf = ctx.mul_add2(curve, k, None, one, g)
    k1 = ints_to_be([(v + 1) % (1 << 256) for v in ks], 32)
    w_xy, w_inf = C.mul_mt(curve, k1, None, threads())
    keep = np.array([v + 1 < (1 << 256) for v in ks])
    assert np.array_equal(s_inf[keep], w_inf[keep]) and np.array_equal(s_xy[keep], w_xy[keep])


def test_config3_secp256k1_variable_base_and_verify_1M(ctx):
    import bench
    n = 1 << 20
    h, r, s, pub, expect = bench.cached_signatures(ctx, n, "ellgpu-bench-v1:3:rank0")
    # P*k (GLV): k = the r column, P = the public keys
    xy, inf = ctx.mul_var("secp256k1", r, pub)
    idx = sample_idx(n)
    # (round 6: every 400th key of the bench batch is off the curve -- x + 1 -- and is reported as
    # such, status 2 and a zeroed result, where the oracle computes with it as the reference does)
    off = bench.expected_status(pub) == 2
    assert 1000 < int(off.sum()) < 4000
    assert (inf[off] == elliptic_amd.STATUS_OFF_CURVE).all() and not xy[off].any() and not (inf[~off] == 2).any()
    idx = idx[~off[idx]]
    want, winf = C.mul_mt("secp256k1", r[idx], pub[idx], threads())
    assert np.array_equal(inf[idx], winf) and np.array_equal(xy[idx], want)
    # verify: mask == expected on ALL tuples (and the status array), expected == oracle on the sample
    ok, st = ctx.ecdsa_verify("secp256k1", h, r, s, pub, status=True)
    assert np.array_equal(np.asarray(ok).astype(np.uint8), expect)
    assert np.array_equal(st, bench.expected_status(pub, r, s))
    idx = sample_idx(n)
    wok = C.verify("secp256k1", h[idx], r[idx], s[idx], pub[idx], threads=threads())
    assert np.array_equal(wok, expect[idx])


def test_config3_strong_scaling_shards_both_tunings(ctx, monkeypatch):
    """BASELINE configs[2] sharded 1 -> 8: a GPU's share of the 2^20 batch is 2^20 / N tuples.
    Shards of at most three waves per SIMD run the small-grid tuning of ecdsa_main
    (FnEcdsaMain<.., WIDE>), larger ones the full-grid tuning; both must give the expected mask on
    every shard of every split, and the two tunings the same bytes on the same shard."""
    import bench
    import torch
    from elliptic_amd.sharding import shard_range
    n = 1 << 20
    h, r, s, pub, expect = bench.cached_signatures(ctx, n, "ellgpu-bench-v1:3:rank0")
    dev = torch.device("cuda", 0)
    t = [torch.from_numpy(x).to(dev) for x in (h, r, s, pub)]
    for world in (2, 4, 8, 7):
        for rank in (0, world - 1):
            lo, hi = shard_range(n, rank, world)
            ok = torch.zeros(hi - lo, dtype=torch.uint8, device=dev)
            ctx.ecdsa_verify_dev("secp256k1", *(x[lo:hi] for x in t), ok)
            torch.cuda.synchronize()
            assert np.array_equal(ok.cpu().numpy(), expect[lo:hi]), (world, rank)
    lo, hi = shard_range(n, 3, 8)
    masks, outs = [], []
    # full-grid tuning, small-grid tuning (prep || table -> ladder), small-grid tuning as prep ->
    # ecdsa_main: the overrides are read when a context is created
    for env in ({"ELLGPU_SMALL_GRID": "0"}, {"ELLGPU_SMALL_GRID": str(1 << 30)},
                {"ELLGPU_SMALL_GRID": str(1 << 30), "ELLGPU_SPLIT_VERIFY": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c2 = elliptic_amd.Context(0)
        for k in env:
            monkeypatch.delenv(k)
        ok = torch.zeros(hi - lo, dtype=torch.uint8, device=dev)
        c2.ecdsa_verify_dev("secp256k1", *(x[lo:hi] for x in t), ok)
        torch.cuda.synchronize()
        masks.append(ok.cpu().numpy())
        # P*k on the same shard: identical bytes in both tunings, and the oracle's on a sample
        outs.append(c2.mul_var("secp256k1", r[lo:hi], pub[lo:hi]))
        c2.close()
    assert all(np.array_equal(m, expect[lo:hi]) for m in masks)
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    j = sample_idx(hi - lo, 1024)
    j = j[bench.expected_status(pub[lo:hi][j]) == 0]     # (the bench batch's off-curve keys: status 2, not the oracle's value)
    want, winf = C.mul_mt("secp256k1", r[lo:hi][j], pub[lo:hi][j], threads())
    assert np.array_equal(outs[1][1][j], winf) and np.array_equal(outs[1][0][j], want)
    idx = sample_idx(hi - lo, 2048) + lo
    assert np.array_equal(C.verify("secp256k1", h[idx], r[idx], s[idx], pub[idx], threads=threads()), expect[idx])


def test_config4_ed25519_variable_base_1M(ctx):
    n = 1 << 20
    k = rnd("bl:cfg4:k", n, 32)
    d = rnd("bl:cfg4:d", n, 32)
    k[:, 0] &= 0x1F                                               # < 2^253 (the reference takes any BN)
    d[:, 0] &= 0x0F
    k[0] = 0
    k[1] = 0
    k[1, 31] = 1
    pts, _ = ctx.mul_fixed("ed25519", d)
    out, inf = ctx.mul_var("ed25519", k, pts)
    idx = sample_idx(n)
    # the generator leg and the variable-base leg against the C oracle
    assert np.array_equal(pts[idx], C.ed_mul(d[idx], None, threads()))
    want = C.ed_mul(k[idx], pts[idx], threads())
    got = out[idx].copy()
    # identity: the engine reports it through the flag with zeroed coordinates, the oracle as (0, 1)
    ident = inf[idx] != 0
    assert ident.sum() >= 1
    assert np.array_equal(got[~ident], want[~ident])
    one = np.zeros(64, np.uint8)
    one[63] = 1
    assert (want[ident] == one).all()


def test_config5_p384_variable_base_256K(ctx):
    n = 1 << 18
    k = rnd("bl:cfg5:k", n, 48)
    d = rnd("bl:cfg5:d", n, 48)
    pts, pinf = ctx.mul_fixed("p384", d)
    assert not pinf.any()
    out, inf = ctx.mul_var("p384", k, pts)
    idx = sample_idx(n)
    want, winf = C.mul_mt("p384", k[idx], pts[idx], threads())
    assert np.array_equal(inf[idx], winf) and np.array_equal(out[idx], want)
    wantg, winfg = C.mul_mt("p384", d[idx[:2048]], None, threads())
    assert np.array_equal(pts[idx[:2048]], wantg) and not winfg.any()


def test_x25519_1M(ctx):
    n = 1 << 20
    k = rnd("bl:x:k", n, 32)
    x = rnd("bl:x:x", n, 32)
    x[:, 0] &= 0x7F
    out, inf = ctx.x25519(k, x)
    idx = sample_idx(n)
    want, winf = C.mont_mul(k[idx], x[idx], threads())
    assert np.array_equal(inf[idx], winf)
    keep = winf == 0
    assert np.array_equal(out[idx][keep], want[keep])


def test_p521_large_batch_kernels_vs_oracle(ctx):
    """batches above 65 536 items take the two-waves-per-SIMD instantiation of the p521 ladders
    (engine.h ELL_P521_PAIR_MIN): those kernels directly against the oracle"""
    n = (1 << 17) + 4096
    k = rnd("bl:p521:k", n, 66)
    d = rnd("bl:p521:d", n, 66)
    k[:, 0] &= 1
    d[:, 0] &= 1
    pts, pinf = ctx.mul_fixed("p521", d)
    out, inf = ctx.mul_var("p521", k, pts)
    idx = sample_idx(n, 4096)
    wg, wgi = C.mul_mt("p521", d[idx], None, threads())
    assert np.array_equal(pinf[idx], wgi) and np.array_equal(pts[idx], wg)
    want, winf = C.mul_mt("p521", k[idx], pts[idx], threads())
    assert np.array_equal(inf[idx], winf) and np.array_equal(out[idx], want)
    sm, sinf = ctx.mul_add2("p521", d, None, k, pts)
    w2, w2i = C.mul_add("p521", d[idx[:1024]], None, k[idx[:1024]], pts[idx[:1024]])
    assert np.array_equal(sinf[idx[:1024]], w2i) and np.array_equal(sm[idx[:1024]], w2)


def test_device_group_on_gpu(ctx):
    """ellgpu_group_create on hardware: ndev = 1 and a two-member group (the same GPU twice: two
    contexts driven by two host threads at once) give the single-context results on an uneven
    batch; the group is what `new Engine({devices})` / Context(devices=...) hand to callers"""
    import bench
    n = 300001
    h, r, s, pub, expect = bench.cached_signatures(ctx, 1 << 20, "ellgpu-bench-v1:3:rank0")
    h, r, s, pub, expect = h[:n], r[:n], s[:n], pub[:n], expect[:n]
    for devs in ([0], [0, 0]):
        g = elliptic_amd.Context(devices=devs)
        assert g.group_size() == len(devs)
        ok = g.ecdsa_verify("secp256k1", h, r, s, pub)
        assert np.array_equal(np.asarray(ok).astype(np.uint8), expect)
        xy, inf = g.mul_var("secp256k1", r, pub)
        xy1, inf1 = ctx.mul_var("secp256k1", r, pub)
        assert np.array_equal(xy, xy1) and np.array_equal(inf, inf1)
        fx, fi = g.mul_fixed("secp256k1", s)
        fx1, fi1 = ctx.mul_fixed("secp256k1", s)
        assert np.array_equal(fx, fx1) and np.array_equal(fi, fi1)
        g.close()


def test_device_group_over_distinct_gpus(ctx):
    """A group over DISTINCT ordinals -- every GPU of the box, at least two -- on an uneven batch:
    verify / P*k / G*k / k1*G + k2*P bytes equal the single-context ones, each member built its own
    tables on its own device, and the N-API engine does the same through `new Engine({devices})`.
    Skipped on a one-GPU box (the 8-GPU node of the round-end scaling run executes it)."""
    import shutil
    import subprocess
    import bench
    from elliptic_amd import _lib
    ndev = _lib.load().ellgpu_device_count()
    if ndev < 2:
        pytest.skip("one GPU on this box (ellgpu_device_count() = %d)" % ndev)
    devs = list(range(min(ndev, 8)))
    n = 300001 + len(devs)                               # never a multiple of the group size
    h, r, s, pub, expect = bench.cached_signatures(ctx, 1 << 20, "ellgpu-bench-v1:3:rank0")
    h, r, s, pub, expect = h[:n], r[:n], s[:n], pub[:n], expect[:n]
    pub = pub.copy()
    pub[7::1001, 63] ^= 1                                # keys off the curve: status 2 beside a verdict of 0
    g = elliptic_amd.Context(devices=devs)
    assert g.group_size() == len(devs)
    ok1, st1 = ctx.ecdsa_verify("secp256k1", h, r, s, pub, status=True)
    ok, st = g.ecdsa_verify("secp256k1", h, r, s, pub, status=True)
    assert np.array_equal(ok, ok1) and np.array_equal(st, st1) and (st == 2).sum() > 200
    on = st1 == 0
    assert np.array_equal(ok[on], expect[on])
    for got, want in ((g.mul_var("secp256k1", r, pub), ctx.mul_var("secp256k1", r, pub)),
                      (g.mul_fixed("secp256k1", s), ctx.mul_fixed("secp256k1", s)),
                      (g.mul_add2("secp256k1", s, None, r, pub), ctx.mul_add2("secp256k1", s, None, r, pub))):
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # a small batch too: fewer items than members' waves, and fewer items than members
    for m in (1, len(devs) - 1, len(devs) + 1, 1000):
        assert np.array_equal(g.ecdsa_verify("secp256k1", h[:m], r[:m], s[:m], pub[:m]), ok1[:m]), m
    g.close()
    # every member alone, on its own ordinal, gives the same bytes (replicated tables)
    for d in devs[1:]:
        c = elliptic_amd.Context(d)
        assert np.array_equal(c.ecdsa_verify("secp256k1", h[:4099], r[:4099], s[:4099], pub[:4099]), ok1[:4099]), d
        c.close()
    if shutil.which("node"):
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        js = ("var e=new (require('%s/elliptic_amd/js').Engine)({devices:%s});var c=require('crypto');"
              "var k=c.createHash('sha512').update('grp').digest().slice(0,32);var n=1003;"
              "var ks=Buffer.concat(Array.from({length:n},function(_,i){var b=Buffer.from(k);b.writeUInt32BE(i+1,28);return b;}));"
              "var one=new (require('%s/elliptic_amd/js').Engine)({device:0});"
              "var a=e.mulBatch('secp256k1',ks,null),b=one.mulBatch('secp256k1',ks,null);"
              "var p=e.mulBatch('secp256k1',ks,a.xy),q=one.mulBatch('secp256k1',ks,b.xy);"
              "console.log(JSON.stringify({fixed:a.xy.equals(b.xy)&&a.inf.equals(b.inf),vari:p.xy.equals(q.xy)&&p.inf.equals(q.inf),n:n}));"
              % (root, json.dumps(devs), root))
        from elliptic_amd.js import build as jb
        if jb.build() is not None:
            pr = subprocess.run(["node", "-e", js], capture_output=True, text=True, timeout=600)
            assert pr.returncode == 0, pr.stderr[-2000:]
            res = json.loads(pr.stdout.strip().splitlines()[-1])
            assert res["fixed"] and res["vari"], res


def test_user_defined_curve_large_batch(ctx):
    """brainpoolP256r1 as a user-defined curve (ellgpu_curve_define_short: run-time prime,
    arbitrary a), 2^17 scalar multiplications and 2^16 k1*P1 + k2*P2 with >= 10 240 of each
    compared with the oracle's generic-a path, a second user-defined curve interleaved (the
    parameter block is swapped per call), and (n-1)*P == -P over every item."""
    import parity_checks as PC
    from golden_util import I
    specs = {s["name"]: s for s in PC.custom_curves()}
    sp = specs["brainpoolP256r1"]
    p, n_ord = I(sp["p"]), I(sp["n"])
    cid = ctx.define_short(p, I(sp["a"]), I(sp["b"]))
    s2 = specs["secp192k1"]
    cid2 = ctx.define_short(I(s2["p"]), I(s2["a"]), I(s2["b"]))
    name = C.define_short("custom:brainpoolP256r1", p, I(sp["a"]), I(sp["b"]), n_ord, I(sp["g"]["x"]), I(sp["g"]["y"]))
    name2 = C.define_short("custom:secp192k1", I(s2["p"]), I(s2["a"]), I(s2["b"]), I(s2["n"]), I(s2["g"]["x"]), I(s2["g"]["y"]))
    n = 1 << 17
    g = np.frombuffer(I(sp["g"]["x"]).to_bytes(32, "big") + I(sp["g"]["y"]).to_bytes(32, "big"), np.uint8)
    # points: r_i * G on the device itself (checked through the sample below)
    pts, pinf = ctx.mul_var(cid, rnd("bl:custom:r", n, 32), np.tile(g, (n, 1)))
    assert pinf.sum() == 0
    k = rnd("bl:custom:k", n, 32)
    xy, inf = ctx.mul_var(cid, k, pts)
    # the other curve in between: its block replaces brainpool's in constant memory
    g2 = np.frombuffer(I(s2["g"]["x"]).to_bytes(32, "big") + I(s2["g"]["y"]).to_bytes(32, "big"), np.uint8)
    k2 = rnd("bl:custom:k2", 4096, 32)
    xy2, inf2 = ctx.mul_var(cid2, k2, np.tile(g2, (4096, 1)))
    w2, wi2 = C.mul_mt(name2, k2, np.tile(g2, (4096, 1)), threads())
    assert np.array_equal(inf2, wi2) and np.array_equal(xy2, w2)
    idx = sample_idx(n)
    wp, wpi = C.mul_mt(name, rnd("bl:custom:r", n, 32)[idx], np.tile(g, (len(idx), 1)), threads())
    assert np.array_equal(pts[idx], wp) and wpi.sum() == 0
    want, winf = C.mul_mt(name, k[idx], pts[idx], threads())
    assert np.array_equal(inf[idx], winf) and np.array_equal(xy[idx], want)
    # (n-1) * P == -P for every item
    km1 = np.tile(np.frombuffer((n_ord - 1).to_bytes(32, "big"), np.uint8), (n, 1))
    neg, ninf = ctx.mul_var(cid, km1, pts)
    assert ninf.sum() == 0 and np.array_equal(neg[:, :32], pts[:, :32])
    ysum = [int.from_bytes(neg[i, 32:].tobytes(), "big") + int.from_bytes(pts[i, 32:].tobytes(), "big") for i in range(0, n, 997)]
    assert all(v == p for v in ysum)
    # k1*P1 + k2*P2
    m = 1 << 16
    ka, kb = rnd("bl:custom:ka", m, 32), rnd("bl:custom:kb", m, 32)
    out, oinf = ctx.mul_add2(cid, ka, pts[:m], kb, pts[m:2 * m])
    idx = sample_idx(m)
    want, winf = C.mul_add(name, ka[idx], pts[:m][idx], kb[idx], pts[m:2 * m][idx])
    assert np.array_equal(oinf[idx], winf) and np.array_equal(out[idx], want)


def test_bench_flow_with_four_ranks_on_one_device():
    """`bench.py --gpus 4` as the driver launches it, with all four ranks on THIS device over gloo (a one-GPU
    box cannot give RCCL four devices): weak + strong blocks, the gathered mask compared with the expected
    mask on every rank.  Three runs: the processes race each other for the device, which is what found the
    unordered zero fill in OverlappedGather (round 6: 3 of 16 eight-rank runs failed the mask check)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for run in range(3):
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "3", "--warmup", "2",
                            "--dist-backend", "gloo", "--force-device", "0", "--batch", "65536", "--sustain", "0",
                            "--no-cpu", "--no-configs", "--no-live-counters"], capture_output=True, text=True, timeout=900, cwd=root)
        assert p.returncode == 0 and "PARITY FAILURE" not in p.stderr, (run, p.stdout[-1500:], p.stderr[-3000:])
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == 4 and line["rccl"]["nranks_seen"] == 4, line.get("rccl")
        assert "equals the global batch's expected mask" in line["strong"]["parity"], line["strong"]
        assert len(line["per_rank"]) == 4
