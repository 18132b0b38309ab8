#!/usr/bin/env python3
"""bench.py -- secp256k1 ECDSA verify batch throughput on N MI355X.

Workload (BASELINE.json configs[2], the configuration the metric "EC
scalar-mults/sec (secp256k1 verify batch)" is quoted on): a batch of 2^20
synthetic (hash, r, s, pubkey) tuples per GPU, 1 % of them corrupted, already
resident in HBM; one "step" = one pass of the hot path (ellgpu_ecdsa_verify_dev:
range checks, batched s^-1 mod n, u1*G + u2*Q with GLV, projective x-compare)
over the whole batch.  Weak scaling: every rank owns its own 2^20 tuples, the
only collective is the final gather of the ok-masks (RCCL all_gather).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work per unit, SURVEY.md 8(d): reference field mul+sqr count x
# (2*8^2 + 8) 32-bit MACs; bytes = 160 in + 1 out
MACS_PER_VERIFY = 2236 * 136
BYTES_PER_VERIFY = 161
HBM_PEAK_GBS = 8000.0

SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def xof(seed: str, nbytes: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(nbytes), dtype=np.uint8)


def make_signatures(ctx, n, seed, corrupt_every=100):
    """n synthetic secp256k1 signatures, all distinct keys/nonces, built without
    any modular inversion: pick d, k, s; r = x(kG) mod n; z = s*k - r*d mod n.
    Every `corrupt_every`-th tuple gets one bit flipped in z, r or s.
    Returns (hash, r, s, pub, expected_ok) as numpy arrays."""
    from elliptic_amd import ints_to_be
    N = SECP_N
    raw = xof(seed + ":d", n * 40).reshape(n, 40)
    ds = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in raw]
    raw = xof(seed + ":k", n * 40).reshape(n, 40)
    ks = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in raw]
    raw = xof(seed + ":s", n * 40).reshape(n, 40)
    ss = [int.from_bytes(row.tobytes(), "big") % (N - 1) + 1 for row in raw]
    pub, inf = ctx.mul_fixed("secp256k1", ints_to_be(ds, 32))
    assert not inf.any()
    R, inf = ctx.mul_fixed("secp256k1", ints_to_be(ks, 32))
    assert not inf.any()
    rs, zs = [], []
    for i in range(n):
        r = int.from_bytes(R[i, :32].tobytes(), "big") % N
        rs.append(r)
        zs.append((ss[i] * ks[i] - r * ds[i]) % N)
    h = ints_to_be(zs, 32)
    r = ints_to_be(rs, 32)
    s = ints_to_be(ss, 32)
    ok = np.ones(n, np.uint8)
    ok[np.array(rs) == 0] = 0
    idx = np.arange(0, n, corrupt_every)
    for j, i in enumerate(idx):
        which = (h, r, s)[j % 3]
        which[i, 31 - (j % 8)] ^= 1 << (j % 7)
        ok[i] = 0
    return h, r, s, pub, ok


def cpu_baseline(h, r, s, pub, ok, budget_s=15.0):
    """The oracle (CPU restatement of the reference's algorithm, oracle/) timed on
    this host on a bounded sample of the same tuples.  Checker only."""
    built = None
    try:
        from oracle import c_oracle
        built = c_oracle.load()
    except Exception:
        built = None
    from elliptic_amd import be_to_ints
    if built is not None:
        return c_oracle.bench_verify(built, h, r, s, pub, ok, budget_s)
    from oracle import ec_oracle as O
    cur = O.get_curve("secp256k1")
    m = min(len(ok), 4000)
    zs, rs, ss = be_to_ints(h[:m]), be_to_ints(r[:m]), be_to_ints(s[:m])
    qx, qy = be_to_ints(pub[:m, :32]), be_to_ints(pub[:m, 32:])
    t0 = time.perf_counter()
    done = 0
    for i in range(m):
        got = O.ecdsa_verify(cur, zs[i], 32, rs[i], ss[i], cur.point(qx[i], qy[i]))
        assert got == bool(ok[i]), "oracle disagrees with the expected mask at %d" % i
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": done / dt, "unit": "verifies/s", "cores": 1, "kind": "port",
            "sample": "first %d tuples of the rank-0 batch, oracle/ec_oracle.py (python ints), 1 thread" % done}


def reference_js_from_profiles():
    """The reference's own pure-JS path, timed by tools/bench_reference_js.js in the build
    container (the GPU box holds no copy of the reference): quoted, never re-measured here."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*reference_js_cpu.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except ValueError:
        return None
    return {"verifies_per_s_per_core": d.get("verify_random_per_s"), "cpu": d.get("cpu"), "node": d.get("node"),
            "where": d.get("where"), "source": os.path.relpath(files[-1], ROOT)}


def traffic_from_profiles():
    """HBM bytes per ecdsa_main launch from the newest committed rocprofv3 PMC summary
    (profiles/*pmc_fetch_write*.txt; FETCH_SIZE / WRITE_SIZE are reported in KiB and were
    collected in separate --pmc passes).  The gfx950 x2 correction of FETCH_SIZE applies to
    wide coalesced streams; this kernel's fetches are 16-byte-per-lane table gathers, for which
    the factor is uncalibrated, so both figures are returned."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_fetch_write*.txt")))
    if not files:
        return None
    txt = open(files[-1]).read()
    f = re.search(r"FnEcdsaMain<CvSecp256k1>>\s+FETCH_SIZE\s+\d+\s+n=\d+\s+per_dispatch=(\d+)", txt)
    w = re.search(r"FnEcdsaMain<CvSecp256k1>>\s+WRITE_SIZE\s+\d+\s+n=\d+\s+per_dispatch=(\d+)", txt)
    if not f or not w:
        return None
    fetch, write = int(f.group(1)) * 1024, int(w.group(1)) * 1024
    return {"bytes_per_launch": fetch + write, "bytes_per_launch_fetch_x2": 2 * fetch + write,
            "fetch_bytes": fetch, "write_bytes": write, "source": os.path.relpath(files[-1], ROOT),
            "note": "per 2^20-tuple launch; dominated by the per-lane window tables (1 KiB written, "
                    "~4.2 KiB gathered per verify), see DESIGN.md section 3"}


def issue_from_profiles(kernel_ms, n):
    """Instruction-issue accounting of ecdsa_main from the newest committed rocprofv3 PMC
    summaries (profiles/*pmc_sq_a*.txt: SQ_INSTS_VALU; *pmc_sq_b*.txt: SQ_INSTS_VALU_INT64 /
    _INT32), priced at the issue times measured by tools/microbench/valu_patterns.hip
    (profiles/*valu_patterns.log: 4.5 cycles per wavefront for v_mad_u64_u32, 4.2 for every
    carry-consuming / VOP3 integer op, 2.5 for plain VOP1/VOP2 ops).  `frac` = issue cycles the
    instruction mix needs / cycles the kernel took per SIMD: the honest utilisation figure for a
    kernel in which a carry costs as much as a multiply."""
    import glob
    import re

    def counter(pattern, name):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
        if not files:
            return None, None
        txt = open(files[-1]).read()
        m = re.search(r"FnEcdsaMain<CvSecp256k1>>\s+" + name + r"\s+(\d+)\s+n=(\d+)", txt)
        c = re.search(r"FnEcdsaMain<CvSecp256k1>>\s+(\d+)\s+\d+\s+\d+\s+[\d.]+", txt)   # kernel-stats row: calls
        if not m or not c:
            return None, None
        return int(m.group(1)) / int(c.group(1)), os.path.relpath(files[-1], ROOT)

    valu, src_a = counter("*pmc_sq_a*.txt", "SQ_INSTS_VALU")
    i64, src_b = counter("*pmc_sq_b*.txt", "SQ_INSTS_VALU_INT64")
    i32, _ = counter("*pmc_sq_b*.txt", "SQ_INSTS_VALU_INT32")
    if not valu or not i64 or not i32:
        return None
    waves = (1 << 20) / 64.0                      # the profiled launches are 2^20-tuple launches
    valu, i64, i32 = valu / waves, i64 / waves, i32 / waves
    other = max(valu - i64 - i32, 0.0)
    need = 4.5 * i64 + 4.2 * i32 + 2.5 * other    # issue cycles per wavefront
    simds = 256 * 4
    clock_khz = 2.4e6
    have = kernel_ms * clock_khz / ((n / 64.0) / simds)
    return {"valu_insts_per_unit": valu, "mad_u64_per_unit": i64, "carry_int32_per_unit": i32,
            "other_per_unit": other, "issue_cycles_per_wave": need, "elapsed_cycles_per_wave_slot": have,
            "frac": need / have if have else None, "clock_ghz": 2.4,
            "sources": [src_a, src_b, "profiles/r01_valu_patterns.log"]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1 << 20, help="tuples per GPU")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (self-test of the N>1 flow)")
    ap.add_argument("--force-device", type=int, default=None,
                    help="self-test only: every rank uses this device (lets the N>1 flow run on a 1-GPU box)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    import elliptic_amd
    ctx = elliptic_amd.Context(local_rank)
    n = args.batch
    h, r, s, pub, expect = make_signatures(ctx, n, "ellgpu-bench-v1:3:rank%d" % rank)
    dev = torch.device("cuda", local_rank)
    dh, dr, dsg, dq = (torch.from_numpy(x).to(dev) for x in (h, r, s, pub))
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    gathered = [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(world)] if world > 1 else None
    ctx.reserve("secp256k1", n)

    def step():
        ctx.ecdsa_verify_dev("secp256k1", dh, dr, dsg, dq, dok)
        if world > 1:
            if args.dist_backend == "nccl":
                dist.all_gather(gathered, dok)      # the final gather, RCCL over xGMI
            else:                                   # gloo self-test: host tensors
                hg = [torch.zeros(n, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(hg, dok.cpu())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # parity at full size: the mask must equal the expected one exactly
    got = dok.cpu().numpy()
    if not np.array_equal(got, expect):
        bad = int((got != expect).sum())
        raise SystemExit("PARITY FAILURE: %d of %d verify results differ from the expected mask" % (bad, n))

    ctx.set_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    timing = ctx.get_timing()
    ctx.set_timing(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total = n * world * args.steps
        value = total / dt
        cnt, main_ms = timing.get("ecdsa_main", (0, 0.0))
        pcnt, prep_ms = timing.get("ecdsa_prep", (0, 0.0))
        k_ms = main_ms / max(cnt, 1)
        # integer-VALU peak: dependency-free v_mad_u64_u32 stream on every CU
        # (best of five short runs: a single one moves by +-3 % with the clock state)
        peak_gmacs = 0.0
        for _ in range(5):
            ms, ops = ctx.probe_valu(0, 256 * 8 * 4, 4096)
            peak_gmacs = max(peak_gmacs, ops / (ms * 1e-3) / 1e9)
        ach_gmacs = n * MACS_PER_VERIFY / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
        ach_gbs = n * BYTES_PER_VERIFY / (k_ms * 1e-3) / 1e9 if k_ms else 0.0
        out = {
            "metric": "secp256k1 ECDSA verify batch throughput (1 verify = 1 double-scalar mult u1*G+u2*Q)",
            "value": value,
            "unit": "verifies/s",
            "scalar_mults_per_s": 2 * value,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": "secp256k1 ECDSA verify (GLV variable-base + fixed-base comb), "
                                   "batch=2^20 tuples per GPU resident in HBM, 1% corrupted"
                                   if n == 1 << 20 else "secp256k1 ECDSA verify, batch=%d per GPU" % n,
                       "batch_per_gpu": n, "parallelism": "shard%d" % world,
                       "parity": "ok-mask == expected mask on all %d tuples" % n},
            "roofline": {
                "bound": "valu_int32_mac",
                "kernel": "ecdsa_main",
                "achieved": ach_gmacs, "peak": peak_gmacs, "unit": "GMAC/s",
                "frac": ach_gmacs / peak_gmacs if peak_gmacs else None,
                "kernel_ms": k_ms, "prep_kernel_ms": prep_ms / max(pcnt, 1),
                "alg_macs_per_unit": MACS_PER_VERIFY,
                "hbm": {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": ach_gbs / HBM_PEAK_GBS, "alg_bytes_per_unit": BYTES_PER_VERIFY},
                "traffic": traffic_from_profiles() if n == 1 << 20 else None,
                "issue": issue_from_profiles(k_ms, n) if n == 1 << 20 else None,
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(h, r, s, pub, expect)
            out["cpu_baseline"]["reference_js"] = reference_js_from_profiles()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
