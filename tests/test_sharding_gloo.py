"""The N>1 path on CPU: world_size-2 gloo process group, each rank verifies its
contiguous shard (through the hostsim build of the device code, since this
container has no GPU) and the masks are gathered -- the same
elliptic_amd.sharding code bench.py uses with backend "nccl" on MI355X."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

from elliptic_amd.sharding import shard_range  # noqa: E402


def test_shard_ranges_cover_exactly():
    for n in (0, 1, 2, 7, 8, 9, 1000, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, lib_path, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import elliptic_amd
        from elliptic_amd import _lib
        from elliptic_amd.sharding import ShardedVerifier
        from golden_util import I, verify_cases
        from elliptic_amd import ints_to_be
        lib = _lib.load(lib_path, optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing", "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
        ctx = elliptic_amd.Context(0, lib_path=lib)
        cs = [c for c in verify_cases("secp256k1") if len(c["z"]) == 64 and "msgBitLength" not in c][:n]
        h = ints_to_be([I(c["z"]) for c in cs], 32)
        r = ints_to_be([I(c["r"]) for c in cs], 32)
        s = ints_to_be([I(c["s"]) for c in cs], 32)
        pub = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], 32), ints_to_be([I(c["qy"]) for c in cs], 32)], axis=1)
        sv = ShardedVerifier(ctx, "secp256k1", dist=dist)
        ok = sv.verify(h, r, s, pub).numpy()
        want = np.array([1 if c["ok"] else 0 for c in cs], np.uint8)
        q.put((rank, bool(np.array_equal(ok, want)), len(cs)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [37])
def test_world2_sharded_verify_matches_reference(n):
    from hostsim.build import build as build_hostsim
    lib_path = build_hostsim()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_worker, args=(r, 2, port, lib_path, n, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1]
    assert all(r[1] for r in res) and all(r[2] == n for r in res)


def _mul_worker(rank, world, port, lib_path, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import elliptic_amd
        from elliptic_amd import _lib
        from elliptic_amd.sharding import ShardedMul
        from golden_util import I, mul_cases, res_xy
        from elliptic_amd import ints_to_be
        lib = _lib.load(lib_path, optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing", "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
        ctx = elliptic_amd.Context(0, lib_path=lib)
        good = True
        for curve, B in (("secp256k1", 32), ("p384", 48)):
            cs = [c for c in mul_cases(curve) if c["op"] == "var"][:23]      # odd count: uneven shards
            k = ints_to_be([I(c["k"]) for c in cs], B)
            pts = np.concatenate([ints_to_be([I(c["px"]) for c in cs], B), ints_to_be([I(c["py"]) for c in cs], B)], axis=1)
            xy, inf = ShardedMul(ctx, curve, dist=dist).mul(k, pts)
            xy, inf = xy.numpy(), inf.numpy()
            for i, c in enumerate(cs):
                want = res_xy(c["r"])
                good = good and ((inf[i] == 1) if want is None else
                                 (inf[i] == 0 and xy[i].tobytes() == want[0].to_bytes(B, "big") + want[1].to_bytes(B, "big")))
            fx = [c for c in mul_cases(curve) if c["op"] == "fixed"][:9]
            xy, inf = ShardedMul(ctx, curve, dist=dist).mul(ints_to_be([I(c["k"]) for c in fx], B))
            xy, inf = xy.numpy(), inf.numpy()
            for i, c in enumerate(fx):
                want = res_xy(c["r"])
                good = good and ((inf[i] == 1) if want is None else
                                 (inf[i] == 0 and xy[i].tobytes() == want[0].to_bytes(B, "big") + want[1].to_bytes(B, "big")))
        q.put((rank, bool(good)))
    finally:
        dist.destroy_process_group()


def test_world2_sharded_mul_gathers_points():
    """the point-output gather: every rank ends with the whole batch's affine results"""
    from hostsim.build import build as build_hostsim
    lib_path = build_hostsim()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_mul_worker, args=(r, 2, port, lib_path, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res)


def _overlap_worker(rank, world, port, lib_path, n, steps, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import elliptic_amd
        from elliptic_amd import _lib, ints_to_be
        from elliptic_amd.sharding import OverlappedGather, shard_range
        from golden_util import I, load
        lib = _lib.load(lib_path, optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing", "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
        ctx = elliptic_amd.Context(0, lib_path=lib)
        # the global batch: verify tuples of the reference, off-curve keys among them
        cs = ([c for c in load("verify_secp256k1.json") if len(c["z"]) == 64] +
              [c for c in load("offcurve_secp256k1.json") if c["op"] == "verify"])[:n]
        assert len(cs) == n
        h = ints_to_be([I(c["z"]) for c in cs], 32)
        r = ints_to_be([I(c["r"]) for c in cs], 32)
        s = ints_to_be([I(c["s"]) for c in cs], 32)
        pub = np.concatenate([ints_to_be([I(c["qx"]) for c in cs], 32), ints_to_be([I(c["qy"]) for c in cs], 32)], axis=1)
        cur_n = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
        # a key that is not on the curve (r, s in range): verdict 0 -- the gathered array is a MASK,
        # strictly 0 / 1 (the domain status is a separate, optional array of the C ABI)
        off = np.array([c.get("on") is False and 0 < I(c["r"]) < cur_n and 0 < I(c["s"]) < cur_n for c in cs])
        want = np.array([(0 if off[i] else (1 if c["ok"] else 0)) for i, c in enumerate(cs)], np.uint8)
        lo, hi = shard_range(n, rank, world)
        og = OverlappedGather(n, dist, torch.device("cpu"))
        good = True
        for step in range(steps):
            out = og.begin()
            # (the hostsim build has no device pointers: compute into numpy, copy into the buffer)
            out.copy_(torch.from_numpy(np.ascontiguousarray(ctx.ecdsa_verify("secp256k1", h[lo:hi], r[lo:hi], s[lo:hi], pub[lo:hi]))))
            if step == 2:                                   # a step whose results differ: buffers must not mix
                out.zero_()
            og.submit()
        og.drain()
        last, prev = (steps - 1) & 1, (steps - 2) & 1
        good = good and np.array_equal(og.result(last).numpy(), want) and np.array_equal(og.result(prev).numpy(), want)
        good = good and int(og.result(last).max()) <= 1
        q.put((rank, bool(good), int(off.sum())))
    finally:
        dist.destroy_process_group()


def test_world2_overlapped_gather_of_verify_masks():
    """bench.py's strong-scaling loop: step i's gather (async all_gather_into_tensor, two
    alternating buffers) overlaps step i + 1's verifies; uneven shards; off-curve keys are 0 in
    the gathered mask (never "accept")"""
    from hostsim.build import build as build_hostsim
    lib_path = build_hostsim()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = _free_port()
    procs = [ctxm.Process(target=_overlap_worker, args=(r, 2, port, lib_path, 91, 7, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(r[0] for r in res) == [0, 1] and all(r[1] for r in res)
    assert all(r[2] >= 10 for r in res)                     # the batch did contain off-curve keys
