#!/usr/bin/env python3
"""bench.py -- secp256k1 ECDSA verify batch throughput on N MI355X.

Workload (BASELINE.json configs[2], the configuration the metric "EC
scalar-mults/sec (secp256k1 verify batch)" is quoted on): a batch of 2^20
synthetic (hash, r, s, pubkey) tuples per GPU, 1 % of them corrupted, already
resident in HBM; one "step" = one pass of the hot path (ellgpu_ecdsa_verify_dev:
range checks, batched s^-1 mod n, u1*G + u2*Q with GLV, projective x-compare)
over the whole batch.  Weak scaling (default): every rank owns its own 2^20 tuples, the
only collective is the final gather of the ok-masks (RCCL all_gather).  `--scaling strong`
= BASELINE configs[2] as written: ONE global batch of 2^20 tuples cut into contiguous shards
(elliptic_amd.sharding.shard_range), every rank verifies its shard, the masks are gathered.

    python bench.py --gpus 1 --steps 120 --warmup 10
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ... [--scaling strong]
    python bench.py --gpus N ...        # plain python: re-launches ITSELF under torch.distributed.run

At N > 1 one invocation measures BOTH forms: `value` is the weak-scaling figure (2^20 tuples on
every GPU -- what the N = 1, 2, 4, 8 runs of the contract compare), and `strong` carries BASELINE
configs[2] as written (ONE batch of 2^20 tuples cut into N shards, masks gathered, gathered mask
checked on every rank) timed in the same process group right after it; `rccl` records the backend
and the number of ranks the collective actually saw.

At N = 1 the roofline's instruction counts and HBM bytes come from rocprofv3 PMC passes made
by this very run on this very box (`live_counters`; a 2-step child of this script under
`rocprofv3 --kernel-trace --pmc ...`, separate passes); if rocprofv3 is unavailable the
committed profiles/*kernel_counters.json is used, and only if its source digest equals the
digest compiled into the loaded library -- otherwise `frac` is null and says why.

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
At N = 1 the same line also carries `configs` (BASELINE configs #2, #4, #5 and
the PCIe-inclusive host-buffer figure, each measured in this run) and
`cpu_baseline` (the reference's own JavaScript timed under Node on this host).
"""
import argparse
import contextlib
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

# algorithmic work per unit, SURVEY.md 8(d): reference field mul+sqr count x (2 L^2 + L)
# 32-bit MACs; bytes = in + out at the C ABI
ALG = {
    "verify": {"macs": 2236 * 136, "bytes": 161},                 # secp256k1 ECDSA verify
    "secp256k1_fixed": {"macs": 787 * 136, "bytes": 97},          # config #2
    "secp256k1_var": {"macs": 1656 * 136, "bytes": 161},
    "ed25519_var": {"macs": 2772 * 136, "bytes": 160},            # config #4
    "p384_var": {"macs": 4276 * 300, "bytes": 241},               # config #5
}
MACS_PER_VERIFY = ALG["verify"]["macs"]
BYTES_PER_VERIFY = ALG["verify"]["bytes"]
HBM_PEAK_GBS = 8000.0
N_SIMD = 256 * 4
ORACLE_SAMPLE = 10240                 # tuples checked against the oracle in every run

SECP_N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
SECP_P = (1 << 256) - (1 << 32) - 977


def xof(seed: str, nbytes: int) -> np.ndarray:
    return np.frombuffer(hashlib.shake_256(seed.encode()).digest(nbytes), dtype=np.uint8)


def make_signatures(ctx, n, seed, corrupt_every=100, corrupt_q=False):
    """n synthetic secp256k1 signatures, all distinct keys/nonces, built without
    any modular inversion: pick d, k, s; r = x(kG) mod n; z = s*k - r*d mod n.
    Every `corrupt_every`-th tuple is corrupted ("flip a bit of r, s, z or Q"): in
    turn one bit of z, of r, of s, and the key Q -- alternately x + 1 (a point that is NOT on the
    curve: the engine reports status 2 beside a verdict of 0) and the next tuple's key (on the
    curve, the wrong key) when corrupt_q is set -- the bench's own batches (cached_signatures) set
    it; with corrupt_q off every fourth corrupted tuple takes a bit flip in z, r or s like the
    others.  Returns (hash, r, s, pub, expected_ok) as numpy arrays; expected_status(pub) gives the
    status array that goes with them.  (The public keys and the
    nonce points come from the engine's own fixed-base kernel; bench.py checks a sample of
    the tuples with the oracle in every run, so a wrong comb table cannot hide.)"""
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
    pub = np.array(pub, copy=True)
    for j, i in enumerate(idx):
        if j % 4 == 3 and corrupt_q:
            if (j // 4) % 2 == 0:
                x = (int.from_bytes(pub[i, :32].tobytes(), "big") + 1) % SECP_P
                pub[i, :32] = np.frombuffer(x.to_bytes(32, "big"), np.uint8)
            else:
                pub[i] = pub[(i + 1) % n]
        else:
            which = (h, r, s)[j % 3]
            which[i, 31 - (j % 8)] ^= 1 << (j % 7)
        ok[i] = 0
    return h, r, s, pub, ok


def expected_status(pub, r=None, s=None):
    """the status array ellgpu_ecdsa_verify reports beside the verdicts: 2 (ELLGPU_STATUS_OFF_CURVE)
    where the key does not satisfy y^2 = x^3 + 7 (and r, s are in range), else 0 -- computed here
    with Python integers, independently of the device"""
    x = [int.from_bytes(row[:32].tobytes(), "big") for row in pub]
    y = [int.from_bytes(row[32:].tobytes(), "big") for row in pub]
    st = np.array([2 if (yy * yy - xx * xx * xx - 7) % SECP_P else 0 for xx, yy in zip(x, y)], np.uint8)
    if r is not None and s is not None:
        for i in np.nonzero(st)[0]:
            ri, si = int.from_bytes(r[i].tobytes(), "big"), int.from_bytes(s[i].tobytes(), "big")
            if not (0 < ri < SECP_N and 0 < si < SECP_N):
                st[i] = 0
    return st


def cached_signatures(ctx, n, seed):
    """make_signatures with a /tmp cache (the generator is ~10 s of Python big-int work per 2^20
    tuples; the profiling passes call bench.py several times on one box)"""
    key = hashlib.sha256(("%s:%d:v3" % (seed, n)).encode()).hexdigest()[:16]
    path = os.path.join(tempfile.gettempdir(), "ellgpu_bench_%s.npz" % key)
    if os.path.exists(path):
        try:
            d = np.load(path)
            return d["h"], d["r"], d["s"], d["pub"], d["ok"]
        except Exception:
            pass
    out = make_signatures(ctx, n, seed, corrupt_q=True)
    try:
        np.savez(path + ".tmp.npz", h=out[0], r=out[1], s=out[2], pub=out[3], ok=out[4])
        os.replace(path + ".tmp.npz", path)
    except OSError:
        pass
    return out


# ---- the checker ------------------------------------------------------------------------
def oracle_check(h, r, s, pub, expect, m=ORACLE_SAMPLE):
    """first m tuples of the batch through the oracle (oracle/ec_oracle.c on the host threads;
    the pure-Python restatement on a smaller sample if the C build is unavailable): the
    expected mask -- which the GPU mask is compared with on ALL tuples -- must be the oracle's"""
    m = min(m, len(expect))
    try:
        from oracle import c_oracle
        c_oracle.load()
        ok = c_oracle.verify("secp256k1", h[:m], r[:m], s[:m], pub[:m], threads=c_oracle._usable_cpus())
        what = "oracle/ec_oracle.c"
    except Exception:
        from oracle import ec_oracle as O
        from elliptic_amd import be_to_ints
        m = min(m, 400)
        cur = O.get_curve("secp256k1")
        zs, rs, ss = be_to_ints(h[:m]), be_to_ints(r[:m]), be_to_ints(s[:m])
        qx, qy = be_to_ints(pub[:m, :32]), be_to_ints(pub[:m, 32:])
        ok = np.array([O.ecdsa_verify(cur, zs[i], 32, rs[i], ss[i], cur.point(qx[i], qy[i])) for i in range(m)],
                      np.uint8)
        what = "oracle/ec_oracle.py"
    if not np.array_equal(ok.astype(np.uint8), expect[:m].astype(np.uint8)):
        raise SystemExit("PARITY FAILURE: the oracle disagrees with the expected mask on %d of the first %d tuples"
                         % (int((ok.astype(np.uint8) != expect[:m]).sum()), m))
    return {"tuples": int(m), "by": what}


def reference_js_baseline(h, r, s, pub, expect, budget_s=8.0):
    """The reference's own pure-JS path (oracle/_ref = indutny/elliptic's bundle, copied by
    oracle/make_ref.py) under Node on THIS host: one process for the per-core figure, then one
    process per usable core over disjoint slices of the rank-0 batch (SURVEY.md 8d row 1 /
    benchmarks/index.js:106-109).  None when Node or the copy is missing."""
    import shutil
    from oracle import make_ref, c_oracle
    ref = make_ref.present() or make_ref.build()
    node = shutil.which("node")
    if ref is None or node is None:
        return None
    cores = c_oracle._usable_cpus()
    per = 1536                                       # tuples per process (>= budget * ~150/s)
    need = per * (cores + 1)
    m = min(len(expect), need)
    rec = np.concatenate([h[:m], r[:m], s[:m], pub[:m], expect[:m].reshape(-1, 1).astype(np.uint8)], axis=1)
    assert rec.shape[1] == 161
    path = os.path.join(tempfile.gettempdir(), "ellgpu_bench_tuples_%d.bin" % os.getpid())
    rec.tofile(path)
    script = os.path.join(ROOT, "tools", "bench_reference_verify.js")
    env = dict(os.environ, ELLIPTIC_REFERENCE=ref)

    def run(first, count, secs, extra=()):
        return subprocess.Popen([node, script, path, str(first), str(count), str(secs)] + list(extra), env=env,
                                stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)

    def collect(ps):
        outs = []
        for p in ps:
            so, se = p.communicate(timeout=300)
            if p.returncode != 0:
                raise RuntimeError("bench_reference_verify.js failed: " + se[-400:])
            outs.append(json.loads(so.strip().splitlines()[-1]))
        return outs
    try:
        one = collect([run(0, min(per, m), budget_s * 0.4)])[0]
        fixed = collect([run(0, 1, 2.0, ["fixed"])])[0]
        slices = []
        for c in range(cores):
            first = per * (c + 1)
            if first + 64 > m:
                break
            slices.append((first, min(per, m - first)))
        many = collect([run(f, c, budget_s * 0.6) for f, c in slices]) if slices else []
    finally:
        try:
            os.remove(path)
        except OSError:
            pass
    bad = one["mismatches"] + sum(o["mismatches"] for o in many)
    if bad:
        raise SystemExit("PARITY FAILURE: the reference (Node) disagrees with the expected mask on %d tuples" % bad)
    total = sum(o["per_s"] for o in many) if many else one["per_s"]
    checked = one["done"] + sum(o["done"] for o in many)
    return {"value": total, "unit": "verifies/s", "cores": len(many) if many else 1, "kind": "reference",
            "single_core_value": one["per_s"], "single_core_fixed_tuple": fixed["per_s"],
            "cpu": one["cpu"], "host_logical_cpus": one["logical_cpus"], "usable_cpus": cores,
            "node": one["node"], "reference": "indutny/elliptic %s, dist/elliptic.js (bn.js 4.11.9) via oracle/_ref"
                                              % one["version"],
            "where": "this host (the GPU box when bench.py runs there)",
            "tuples_verified_by_reference": checked,
            "sample": "ec.verify(msg, Signature, KeyPair) as benchmarks/index.js:106-109 calls it, on tuples "
                      "%d..%d of the rank-0 batch: %d Node processes x <= %d tuples, %.1f s each (sum of the "
                      "per-process rates), after 1 process on the first %d; every verdict equals the expected mask"
                      % (per, per * (len(many) + 1), len(many), per, budget_s * 0.6, min(per, m))}


def port_baseline(h, r, s, pub, expect, budget_s=6.0):
    """oracle/ec_oracle.c (C port of the reference's algorithm) on the host threads: the 'fair
    native CPU' line beside the reference's JavaScript"""
    try:
        from oracle import c_oracle
        return c_oracle.bench_verify(c_oracle.load(), h, r, s, pub, expect, budget_s)
    except Exception as e:       # pragma: no cover
        return {"error": str(e)}


# ---- committed profile summaries ------------------------------------------------------------
def _latest(pattern):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return files[-1] if files else None


def kernel_counters():
    """profiles/rNN_kernel_counters.json (tools/refresh_profiles.py: rocprofv3 PMC passes of this
    very command): VALU instructions by class per unit for every benchmarked kernel, VALU-busy,
    HBM traffic, and the source digest of the library they were taken from"""
    f = _latest("*kernel_counters.json")
    if not f:
        return None, None
    try:
        return json.load(open(f)), os.path.relpath(f, ROOT)
    except ValueError:
        return None, None


_STATIC_MIX = None


def static_mix():
    """{kernel key: multiply / other-64-bit counts per unit} from profiles/*static_op_mix.json
    (tools/static_mix.py), only if it was made from the sources of the loaded library"""
    global _STATIC_MIX
    if _STATIC_MIX is None:
        _STATIC_MIX = {}
        f = _latest("*static_op_mix.json")
        try:
            d = json.load(open(f)) if f else None
        except ValueError:
            d = None
        if d and d.get("source_digest") == lib_digest():
            for curve, key in (("p384", "mul_var<p384>"), ("p256", "mul_var<p256>")):
                m = d["curves"].get(curve, {}).get("mul_var_model")
                if m:
                    _STATIC_MIX[key] = dict(m, source=os.path.relpath(f, ROOT))
    return _STATIC_MIX


def _tool(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location("ellgpu_tool_" + name, os.path.join(ROOT, "tools", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


LIVE_PASSES = (
    ("sqa", ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_BUSY_CYCLES",
             "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "GRBM_GUI_ACTIVE"]),
    ("sqb", ["SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD",
             "SQ_INSTS_VMEM_WR", "SQ_WAVES", "SQ_ACTIVE_INST_ANY"]),
    ("fwa", ["FETCH_SIZE"]),
    ("fwb", ["WRITE_SIZE"]),
)


def live_counters(batch, per_pass_timeout=90, keep=None, total_budget=240):
    """The roofline's instruction counts and HBM bytes, measured by THIS run: one rocprofv3 PMC
    pass per counter group (--kernel-trace + --pmc only, MI355X_MICROARCH.md's recipe) over a
    2-step child of this script -- the same binaries, box and inputs as the timed region --
    distilled by tools/refresh_profiles.distil into the structure of profiles/*kernel_counters.json.
    -> (counters | None, source string, note)"""
    import glob
    import io
    import contextlib
    import shutil
    rp = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if rp is None:
        return None, None, "rocprofv3 not found"
    tmp = os.path.abspath(keep) if keep else tempfile.mkdtemp(prefix="ellgpu_pmc_")
    os.makedirs(tmp, exist_ok=True)
    summ = _tool("rocprof_summary")
    child = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu",
             "--no-live-counters", "--sustain", "0", "--batch", str(batch)]
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.perf_counter()
    notes = []
    for tag, ctrs in LIVE_PASSES:
        d = os.path.join(tmp, "prof_" + tag)
        cmd = [rp, "--kernel-trace", "--pmc"] + ctrs + ["-d", d, "-o", tag, "--"] + child
        left = total_budget - (time.perf_counter() - t0)
        if left < 20:                       # never let the profiler stretch the bench beyond minutes
            notes.append("%s: skipped (time budget)" % tag)
            continue
        try:
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True,
                               timeout=min(per_pass_timeout, left))
        except subprocess.TimeoutExpired:
            notes.append("%s: timeout" % tag)
            break                           # a profiler that hangs once is not asked again
        dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
        if p.returncode != 0 or not dbs:
            notes.append("%s: rc=%d %s" % (tag, p.returncode, (p.stderr or "")[-160:].replace("\n", " ")))
            continue
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            summ.main(dbs[0])
        with open(os.path.join(tmp, "rocprof_%s.txt" % tag), "w") as f:
            f.write(buf.getvalue())
        shutil.rmtree(d, ignore_errors=True)
    with open(os.path.join(tmp, "rocprof_fw.txt"), "w") as f:
        for t in ("fwa", "fwb"):
            q = os.path.join(tmp, "rocprof_%s.txt" % t)
            if os.path.exists(q):
                f.write(open(q).read())
    kc = _tool("refresh_profiles").distil(tmp, lib_digest())
    kc["how"] = ("LIVE: rocprofv3 --kernel-trace --pmc, one pass per counter group, over `bench.py --steps 2 "
                 "--warmup 1 --no-cpu` run as a child of this bench.py on this box (%.0f s)" % (time.perf_counter() - t0))
    if kc.get("fetch_calibration") is None:
        # the gather calibration (tools/microbench/gather_calib.hip) is not re-run here; the
        # committed one reads 0.9994 true bytes per counted byte for this access pattern
        old, _ = kernel_counters()
        kc["fetch_calibration"] = (old or {}).get("fetch_calibration")
        kc["fetch_calibration_source"] = "committed profiles/*kernel_counters.json (not re-measured in this run)"
    if keep is None:
        shutil.rmtree(tmp, ignore_errors=True)
    if not kc["kernels"]:
        return None, None, "live PMC passes produced no counters (%s)" % "; ".join(notes)
    return kc, "live rocprofv3 --pmc passes of this run", "; ".join(notes) or None


def lib_digest():
    """the source digest compiled into the libellgpu.so this process loaded (ellgpu_source_digest)"""
    from elliptic_amd import _lib
    try:
        return _lib.load().ellgpu_source_digest().decode()
    except (ImportError, AttributeError, OSError):
        return None


def roofline_block(kernel_key, unit_key, n, kernel_ms, peak_gmads, clock_ghz, counters, csrc):
    """roofline object for one kernel launch over n units that took kernel_ms"""
    alg = ALG[unit_key]
    theo = N_SIMD * 64 / 4.0 * clock_ghz                       # G mad/s if a wave-mad issued every 4 cycles
    out = {"bound": "valu_int32_mac", "kernel": kernel_key, "unit": "G v_mad_u64_u32/s",
           "peak": peak_gmads, "peak_theoretical_4cycle": theo, "kernel_ms": kernel_ms,
           "alg_macs_per_unit": alg["macs"]}
    ach_alg = n * alg["macs"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
    out["frac_reference_alg"] = ach_alg / peak_gmads if peak_gmads else None
    out["achieved_reference_alg"] = ach_alg
    k = (counters or {}).get("kernels", {}).get(kernel_key)
    stale = bool(counters) and counters.get("source_digest") != lib_digest()
    if k and stale:
        # instruction counts of OTHER binaries do not price this run: no fraction at all
        out.update({"achieved": None, "frac": None, "counters_stale": True, "counters_source": csrc,
                    "note": "PMC instruction counts in %s were taken from library digest %s, the loaded "
                            "library is %s: refusing to compute a fraction from them (re-run "
                            "tools/refresh_profiles.sh or let bench.py make its live PMC passes)"
                            % (csrc, counters.get("source_digest"), lib_digest())})
        k = None
    elif k and k.get("mad_u64_per_unit"):
        mads = k["mad_u64_per_unit"]
        # NIST curves: SQ_INSTS_VALU_INT64 also counts the Solinas fold's 64-bit adds / shifts.
        # tools/static_mix.py counts those per field operation from the ISA; the ladder's shape
        # gives the field operations per unit: multiplies = PMC INT64 - that many.
        sm = static_mix().get(kernel_key)
        if sm and sm.get("other_int64_per_unit"):
            out["int64_pmc_per_unit"] = mads
            out["int64_non_multiply_static_per_unit"] = sm["other_int64_per_unit"]
            out["mads_static_model_per_unit"] = sm["mad_u64_u32_per_unit"]
            out["multiply_count_source"] = ("SQ_INSTS_VALU_INT64 minus the fold's 64-bit adds/shifts counted from "
                                            "the ISA x the ladder's field-operation count (%s)" % sm["source"])
            mads = mads - sm["other_int64_per_unit"]
        ach = n * mads / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
        out.update({"achieved": ach, "frac": ach / peak_gmads if peak_gmads else None,
                    "frac_of_theoretical": ach / theo,
                    "mads_issued_per_unit": mads, "valu_insts_per_unit": k.get("valu_per_unit"),
                    "carry_int32_per_unit": k.get("int32_per_unit"),
                    "multiply_share_of_valu_insts": mads / k["valu_per_unit"] if k.get("valu_per_unit") else None,
                    "valu_busy_pmc": k.get("valu_busy"), "counters_source": csrc,
                    "counters_stale": False, "counters_digest": counters.get("source_digest")})
        # issue accounting: a wave64 VALU instruction occupies its 16-lane SIMD for 4 cycles; the
        # per-unit counts are per lane, i.e. per wavefront-instruction stream
        if k.get("valu_per_unit"):
            waves_per_simd = (n / 64.0) / N_SIMD
            elapsed = kernel_ms * 1e-3 * clock_ghz * 1e9
            cpi = elapsed / (waves_per_simd * k["valu_per_unit"]) if waves_per_simd else None
            # two-class issue model (profiles/r04_issue_runs.log): multiplies and
            # carry / select / 64-bit instructions occupy a 4-cycle wave64 slot each, plain 32-bit moves,
            # adds and logic ride behind them for free.  INT64 + INT32 is the PMC's upper bound of the
            # slow class (INT32 also counts the plain adds), so this is the fraction of the kernel's
            # SIMD cycles that such slots account for at the nominal clock (the part sustains ~8 % less)
            slow = (k.get("mad_u64_per_unit") or 0) + (k.get("int32_per_unit") or 0)
            out["issue_slots"] = {"slow_class_insts_per_unit_upper_bound": slow, "cycles_per_slot": 4,
                                  "frac_of_elapsed_cycles_at_nominal_clock":
                                      (waves_per_simd * slow * 4.0 / elapsed) if elapsed and slow else None}
            out["issue"] = {"valu_insts_per_wave": k["valu_per_unit"], "waves_per_simd": waves_per_simd,
                            "elapsed_cycles_at_nominal_clock": elapsed, "cycles_per_valu_inst": cpi,
                            "floor_cycles_per_valu_inst": 4.0,
                            "valu_busy_pct_pmc": (k.get("valu_busy") or {}).get("valu_busy_pct_gfx94x_formula"),
                            "note": "SIMD cycles per VALU wave-instruction at the nominal clock; 4 is the wave64 issue "
                                    "floor of mixed code (runs of plain VOP1/VOP2 issue faster, "
                                    "profiles/*valu_patterns.log): at or below it the VALU pipe never idles"}
    elif not stale:
        out.update({"achieved": None, "frac": None,
                    "note": "no PMC instruction counts for this kernel (neither live nor profiles/*kernel_counters.json)"})
    ach_gbs = n * alg["bytes"] / (kernel_ms * 1e-3) / 1e9 if kernel_ms else 0.0
    out["hbm"] = {"achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                  "alg_bytes_per_unit": alg["bytes"]}
    if k and k.get("fetch_bytes_per_unit") is not None:
        out["traffic"] = {"bytes_per_launch": (k["fetch_bytes_per_unit"] + k["write_bytes_per_unit"]) * n,
                          "fetch_bytes_per_unit": k["fetch_bytes_per_unit"],
                          "write_bytes_per_unit": k["write_bytes_per_unit"],
                          "fetch_calibration": counters.get("fetch_calibration"), "source": csrc}
    else:
        out["traffic"] = None
    return out


# ---- the other BASELINE configs (N = 1 only) ---------------------------------------------------
def run_configs(ctx, dev, peak_gmads, clock_ghz, counters, csrc, h, r, s, pub, expect):
    """BASELINE.json configs #2, #4, #5 (kernel-only, inputs in HBM, a sample of every result
    checked against the oracle) and the PCIe-inclusive form of the headline (host buffers in,
    mask out).  -> list of dicts"""
    import elliptic_amd
    from oracle import c_oracle, ec_oracle as O
    rows = []

    def rnd(seed, n, w):
        return xof(seed, n * w).reshape(n, w).copy()

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        ctx.set_timing(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tm = ctx.get_timing()
        ctx.set_timing(False)
        return dt, {k: v[1] / v[0] for k, v in tm.items()}

    def sample_check(curve, ks, pts, got_xy, got_inf, m=2048):
        """first m results against the oracle (C port for the short curves, python for ed25519)"""
        m = min(m, len(ks))
        if curve == "ed25519":
            want = c_oracle.ed_mul(ks[:m], None if pts is None else pts[:m], c_oracle._usable_cpus())
            ident = got_inf[:m] != 0                      # the engine flags the identity, the oracle returns (0, 1)
            assert np.array_equal(want[~ident], got_xy[:m][~ident]), "ed25519 parity vs oracle"
            assert all(int.from_bytes(w[:32].tobytes(), "big") == 0 and int.from_bytes(w[32:].tobytes(), "big") == 1
                       for w in want[ident]), "ed25519 identity parity"
            return m
        want, winf = c_oracle.mul(curve, ks[:m], None if pts is None else pts[:m])
        assert np.array_equal(winf, got_inf[:m]) and np.array_equal(want, got_xy[:m]), curve + " parity vs oracle"
        return m

    for cfg, curve, n, unit_key, kernel_key, fixed in (
            ("#2 secp256k1 fixed-base G*k, batch 2^20", "secp256k1", 1 << 20, "secp256k1_fixed", "mul_fixed<secp256k1>", True),
            ("#3' secp256k1 variable-base P*k (GLV), batch 2^20", "secp256k1", 1 << 20, "secp256k1_var", "mul_var<secp256k1>", False),
            ("#4 ed25519 variable-base P*k + batch inversion, batch 2^20", "ed25519", 1 << 20, "ed25519_var", "ed_mul_var", False),
            ("#5 p384 variable-base P*k, batch 2^18", "p384", 1 << 18, "p384_var", "mul_var<p384>", False)):
        B = elliptic_amd.FIELD_BYTES[curve]
        ks = rnd("ellgpu-bench-v1:cfg:k:" + curve, n, B)
        if curve == "ed25519":
            ks[:, 0] &= 0x0F                       # scalars < 2^252 < n
        dk = torch.from_numpy(ks).to(dev)
        out = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
        inf = torch.zeros(n, dtype=torch.uint8, device=dev)
        pts_np = None
        if fixed:
            fn = lambda: ctx.mul_fixed_dev(curve, dk, out, inf)          # noqa: E731
        else:
            ds = rnd("ellgpu-bench-v1:cfg:d:" + curve, n, B)
            if curve == "ed25519":
                ds[:, 0] &= 0x0F
            dd = torch.from_numpy(ds).to(dev)
            pts = torch.zeros((n, 2 * B), dtype=torch.uint8, device=dev)
            ctx.mul_fixed_dev(curve, dd, pts, inf)                        # P_i = d_i * G
            torch.cuda.synchronize()
            pts_np = pts.cpu().numpy()
            fn = lambda: ctx.mul_var_dev(curve, dk, pts, out, inf)        # noqa: E731
        reps = max(3, int(0.6 / (n / 100e6)) if fixed else 12)
        dt, kms = timed(fn, min(reps, 40))
        checked = sample_check(curve, ks, pts_np, out.cpu().numpy(), inf.cpu().numpy())
        main_name = max(kms, key=kms.get)
        row = {"config": cfg, "n": n, "items_per_s": n / dt, "ms_per_pass": dt * 1e3, "kernels_ms": kms,
               "passes_in_flight": 1, "dominant_kernel": main_name, "oracle_checked": checked,
               "roofline": roofline_block(kernel_key, unit_key, n, kms[main_name], peak_gmads, clock_ghz, counters, csrc)}
        # the same passes with TWO in flight (alternating streams and result buffers; the context
        # gives every stream its own scratch arena): a batch that is not a whole number of wave
        # rounds -- p384's 2^18 items are 4 waves per SIMD on 3 register slots -- leaves a tail in
        # which the next pass's waves can run
        out2, inf2 = torch.zeros_like(out), torch.zeros_like(inf)
        lanes2 = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        bufs = [(out, inf), (out2, inf2)]

        def fn2(i):
            with torch.cuda.stream(lanes2[i & 1]):
                if fixed:
                    ctx.mul_fixed_dev(curve, dk, bufs[i & 1][0], bufs[i & 1][1])
                else:
                    ctx.mul_var_dev(curve, dk, pts, bufs[i & 1][0], bufs[i & 1][1])
        torch.cuda.synchronize()
        for i in range(2):
            fn2(i)
        torch.cuda.synchronize()
        r2 = 2 * ((min(reps, 40) + 1) // 2)
        t0 = time.perf_counter()
        for i in range(r2):
            fn2(i)
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / r2
        assert torch.equal(out, out2) and torch.equal(inf, inf2), curve + ": two passes in flight gave different bytes"
        row["two_in_flight"] = {"items_per_s": n / dt2, "ms_per_pass": dt2 * 1e3}
        rows.append(row)
        del dk, out, inf, out2, inf2
    # PCIe-inclusive headline: pageable host buffers in, mask out (H2D 160 B + D2H 1 B per tuple)
    n = len(expect)
    got = ctx.ecdsa_verify("secp256k1", h, r, s, pub)
    assert np.array_equal(np.asarray(got).astype(np.uint8), expect.astype(np.uint8)), "host path parity"
    ts = []
    for _ in range(6):
        t0 = time.perf_counter()
        ctx.ecdsa_verify("secp256k1", h, r, s, pub)
        ts.append(time.perf_counter() - t0)
    rows.append({"config": "#3 secp256k1 ECDSA verify through the host-buffer entry point "
                           "(PCIe-inclusive: H2D 160 B + kernels + D2H 1 B per tuple), batch 2^20",
                 "n": n, "items_per_s": n / (sum(ts) / len(ts)), "items_per_s_best": n / min(ts),
                 "ms_per_pass": sum(ts) / len(ts) * 1e3, "note": "never `value`; mask == expected on all tuples"})
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=120)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1 << 20, help="tuples per GPU")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): --batch tuples on EVERY GPU; strong: ONE global batch of --batch tuples "
                         "cut into contiguous shards, one per GPU (BASELINE configs[2] as written)")
    ap.add_argument("--no-live-counters", action="store_true",
                    help="do not make the rocprofv3 PMC passes; price the roofline with the committed counters")
    ap.add_argument("--keep-counters", default=None, help="directory to keep the live PMC summaries in")
    ap.add_argument("--in-flight", type=int, choices=(1, 2), default=2,
                    help="passes in flight: 2 (default) = consecutive steps alternate two streams, each with its "
                         "own result buffer and its own scratch arena of the context, so that step i + 1's "
                         "latency-bound front (s^-1, window tables) runs beside step i's ladder; 1 = one stream")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) | gloo (self-test of the N>1 flow)")
    ap.add_argument("--force-device", type=int, default=None,
                    help="self-test only: every rank uses this device (lets the N>1 flow run on a 1-GPU box)")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1: initialise a 1-rank RCCL group and run the gather through it once")
    ap.add_argument("--sustain", type=float, default=5.0,
                    help="seconds of the same step loop AFTER the timed region, reported as `sustained` (0 = skip)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: become the launcher.  One rank per GPU under
        # torch.distributed.run on 127.0.0.1; rank 0 of the children prints the JSON line on this
        # process's stdout.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run "
                         "--nproc-per-node %d (or as plain `python bench.py --gpus %d`, which launches itself)"
                         % (args.gpus, world, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    if args.force_device is not None:
        local_rank = args.force_device
    torch.cuda.set_device(local_rank)
    dist = None
    rccl_selftest = None
    if world > 1 or args.rccl_selftest:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if world == 1:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.dist_backend)

    import elliptic_amd
    from elliptic_amd.sharding import shard_range, gather_results
    ctx = elliptic_amd.Context(local_rank)
    strong = args.scaling == "strong"
    n_global = args.batch if strong else args.batch * world
    if strong:
        # one global batch (every rank derives the same tuples), rank r owns [lo, hi)
        h, r, s, pub, expect = cached_signatures(ctx, args.batch, "ellgpu-bench-v1:3:rank0")
        checked = oracle_check(h, r, s, pub, expect) if rank == 0 else {"tuples": 0, "by": "rank 0"}
        lo, hi = shard_range(args.batch, rank, world)
        h, r, s, pub, expect = (x[lo:hi] for x in (h, r, s, pub, expect))
    else:
        h, r, s, pub, expect = cached_signatures(ctx, args.batch, "ellgpu-bench-v1:3:rank%d" % rank)
        checked = oracle_check(h, r, s, pub, expect)
    n = len(expect)
    dev = torch.device("cuda", local_rank)
    dh, dr, dsg, dq = (torch.from_numpy(np.ascontiguousarray(x)).to(dev) for x in (h, r, s, pub))
    dok = torch.zeros(n, dtype=torch.uint8, device=dev)
    gathered = None
    ctx.reserve("secp256k1", n)
    # passes in flight: step i runs on stream i % 2 into result buffer i % 2 (the context gives every
    # stream its own scratch arena); all of them complete inside the timed region (synchronize)
    old_strong = strong and world > 1                 # --scaling strong: one synchronous gather per step
    flight = 1 if old_strong else args.in_flight
    lanes = [torch.cuda.Stream(device=dev) for _ in range(flight)] if flight > 1 else [None]
    doks = [dok] + [torch.zeros(n, dtype=torch.uint8, device=dev) for _ in range(flight - 1)]
    step_no = [0]

    def on_lane():
        """-> (context manager of this step's stream, its index)"""
        i = step_no[0] % flight
        step_no[0] += 1
        return (torch.cuda.stream(lanes[i]) if lanes[i] is not None else contextlib.nullcontext()), i
    torch.cuda.synchronize()
    full_mask = None
    # weak scaling at N > 1: the final gather (RCCL over xGMI with backend nccl) runs on the
    # collective's own stream, double-buffered, so that step i's gather overlaps step i + 1's
    # kernels (elliptic_amd.sharding.OverlappedGather); every gather completes inside the timed region
    og_w = None
    if world > 1 and not strong:
        from elliptic_amd.sharding import OverlappedGather
        og_w = OverlappedGather(n * world, dist, dev, gather_device=dev if args.dist_backend == "nccl" else torch.device("cpu"))

    def step():
        nonlocal full_mask
        lane, li = on_lane()
        with lane:
            if og_w is not None:
                ctx.ecdsa_verify_dev("secp256k1", dh, dr, dsg, dq, og_w.begin())
                og_w.submit()
                return
            ctx.ecdsa_verify_dev("secp256k1", dh, dr, dsg, dq, doks[li])
            if world > 1:                               # --scaling strong: uneven shards, padded all_gather, trimmed
                src = dok if args.dist_backend == "nccl" else dok.cpu()
                full_mask = gather_results(src, args.batch, dist)

    for _ in range(max(args.warmup, 2) if og_w is not None else max(args.warmup, flight)):     # every result buffer is written
        step()
    if og_w is not None:
        og_w.drain()
        torch.cuda.synchronize()
        for b in (0, 1):                            # both buffers: this rank's rows, locally and in the gathered result
            if not (np.array_equal(og_w.local[b][:n].cpu().numpy(), expect) and
                    np.array_equal(og_w.result(b)[rank * n:(rank + 1) * n].cpu().numpy(), expect)):
                raise SystemExit("PARITY FAILURE: weak-scaling pass: verify results / gathered rows differ from the expected mask")
        dok.copy_(og_w.local[0][:n])
    torch.cuda.synchronize()
    # parity at full size: the mask must equal the expected one exactly (every result buffer)
    for d in (doks if og_w is None else [dok]):
        got = d.cpu().numpy()
        if not np.array_equal(got, expect):
            bad = int((got != expect).sum())
            raise SystemExit("PARITY FAILURE: %d of %d verify results differ from the expected mask" % (bad, n))
    # ... and the STATUS array beside it: 2 exactly where the (corrupted) key is not on the curve --
    # one pass of the same entry point with out_status, compared with a host-side curve test
    dst = torch.zeros(n, dtype=torch.uint8, device=dev)
    dok_st = torch.zeros(n, dtype=torch.uint8, device=dev)
    ctx.ecdsa_verify_dev("secp256k1", dh, dr, dsg, dq, dok_st, out_status=dst)
    torch.cuda.synchronize()
    want_st = expected_status(pub, r, s)
    got_st = dst.cpu().numpy()
    if not np.array_equal(got_st, want_st) or not np.array_equal(dok_st.cpu().numpy(), expect):
        raise SystemExit("PARITY FAILURE: status array differs from the host-side curve test on %d of %d tuples"
                         % (int((got_st != want_st).sum()), n))
    off_curve_tuples = int((want_st == 2).sum())
    del dst, dok_st
    if strong and world > 1:
        # the gathered mask is the global batch's expected mask on every rank
        _, _, _, _, expect_all = cached_signatures(ctx, args.batch, "ellgpu-bench-v1:3:rank0")
        if not np.array_equal(full_mask.cpu().numpy(), expect_all):
            raise SystemExit("PARITY FAILURE: gathered mask differs from the expected mask of the global batch")
    if world == 1 and dist is not None:
        # the RCCL branch of the N > 1 flow, executed once on a 1-rank group
        t0 = time.perf_counter()
        gathered = gathered or [torch.zeros_like(dok)]
        dist.all_gather(gathered, dok)
        torch.cuda.synchronize()
        rccl_selftest = {"backend": args.dist_backend, "world": 1, "all_gather_ok": bool(torch.equal(gathered[0], dok)),
                         "ms": (time.perf_counter() - t0) * 1e3}

    own_dt = [0.0]

    def timed(fn, steps, finish=None):
        """exactly `steps` calls of fn (+ finish(): whatever they left in flight) between barrier +
        synchronize on both sides -> max over ranks"""
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        if finish is not None:
            finish()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        own_dt[0] = dt                              # this rank's own wall time of the region (see per_rank)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    ctx.set_timing(True)
    dt = timed(step, args.steps, og_w.drain if og_w is not None else None)
    timing = ctx.get_timing()
    ctx.set_timing(False)
    own_dt = [own_dt[0]]                            # (kept: later regions write their own)
    own_main = own_dt[0]
    # with two passes in flight the HIP events around a launch also span the time the kernel
    # shares the device with the other pass's kernels: the kernels BY THEMSELVES are timed in a
    # short one-stream loop outside the timed region
    timing_alone = None
    one_in_flight = None
    alone_steps = min(args.steps, 20)
    if flight > 1 and og_w is None:
        ctx.set_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(alone_steps):
            ctx.ecdsa_verify_dev("secp256k1", dh, dr, dsg, dq, dok)
        torch.cuda.synchronize()
        dt1 = time.perf_counter() - t0
        timing_alone = ctx.get_timing()
        ctx.set_timing(False)
        # the cross-check the line carries with it: with ONE pass in flight a step cannot be shorter
        # than its kernels (kernel_ms + prep_kernel_ms <= one_in_flight.ms_per_step); `value` is the
        # two-in-flight rate, in which the front of pass i + 1 runs beside the ladder of pass i
        one_in_flight = {"value": n * alone_steps / dt1, "unit": "verifies/s", "ms_per_step": dt1 / alone_steps * 1e3,
                         "steps": alone_steps,
                         "how": "this rank's batch in a ONE-stream loop right after the timed region, wall clock "
                                "between synchronisations; never `value`"}

    # SUSTAINED rate (never `value`): the timed region above is --steps passes, a fraction of a second,
    # on a kernel that runs into the part's power limit -- so the same step loop (same streams, same
    # synchronisation, max over ranks) is run again for at least --sustain seconds
    sustained = None
    if args.sustain > 0:
        sus_steps = max(args.steps, int(np.ceil(args.sustain / (dt / args.steps))))
        sdt_s = timed(step, sus_steps, og_w.drain if og_w is not None else None)
        sustained = {"seconds": sdt_s, "steps": sus_steps, "value": n_global * sus_steps / sdt_s, "unit": "verifies/s",
                     "ms_per_step": sdt_s / sus_steps * 1e3,
                     "ratio_to_value": (n_global * sus_steps / sdt_s) / (n_global * args.steps / dt),
                     "how": "the timed region's step loop again (same passes in flight, barrier + synchronize on both "
                            "sides, max over ranks), for at least --sustain seconds"}
        # every result buffer still holds the expected mask
        for d in (doks if og_w is None else []):
            if not np.array_equal(d.cpu().numpy(), expect):
                raise SystemExit("PARITY FAILURE: a result buffer differs from the expected mask after the sustained loop")

    # per-rank view of the timed region (a slow rank must be visible): this rank's own wall time
    # per step and its dominant kernel's mean duration between HIP events
    per_rank = None
    if world > 1:
        cnt_r, main_r = timing.get("ecdsa_main", (0, 0.0))
        mine = torch.tensor([float(rank), own_main / args.steps * 1e3, main_r / max(cnt_r, 1)], dtype=torch.float64,
                            device=dev if args.dist_backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": int(t[0].item()), "ms_per_step": float(t[1].item()), "ecdsa_main_ms": float(t[2].item())} for t in allr]

    rccl = None
    strong_block = None
    if world > 1:
        # how many ranks does the collective really see?
        ones = torch.ones(1, dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(ones)
        rccl = {"backend": args.dist_backend + (" (= RCCL)" if args.dist_backend == "nccl" else ""), "world": world,
                "nranks_seen": int(ones.item()),
                "collectives": "one all_gather_into_tensor of the ok-masks per step (1 B per tuple), issued with "
                               "async_op on the collective's stream and double-buffered, so that it overlaps the "
                               "next step's kernels; max-over-ranks timing and this census are all_reduce calls "
                               "outside the timed steps"}
    if world > 1 and not strong:
        # BASELINE configs[2] as written, in the same process group: ONE batch of --batch tuples
        # (rank 0's, broadcast -- every rank then verifies its contiguous shard), masks gathered,
        # the gathered mask compared with the batch's expected mask on EVERY rank
        nb = args.batch
        if args.dist_backend == "nccl":
            pack = torch.cat([dh, dr, dsg, dq], dim=1) if rank == 0 else torch.zeros((nb, 160), dtype=torch.uint8, device=dev)
            exp_all = torch.from_numpy(expect).to(dev) if rank == 0 else torch.zeros(nb, dtype=torch.uint8, device=dev)
            dist.broadcast(pack, 0)
            dist.broadcast(exp_all, 0)
        else:                                           # gloo self-test: through host memory
            pack = torch.cat([dh, dr, dsg, dq], dim=1).cpu() if rank == 0 else torch.zeros((nb, 160), dtype=torch.uint8)
            exp_all = torch.from_numpy(expect.copy()) if rank == 0 else torch.zeros(nb, dtype=torch.uint8)
            dist.broadcast(pack, 0)
            dist.broadcast(exp_all, 0)
            pack, exp_all = pack.to(dev), exp_all.to(dev)
        lo, hi = shard_range(nb, rank, world)
        sh = pack[lo:hi]
        sh_h, sh_r, sh_s, sh_q = (sh[:, a:b].contiguous() for a, b in ((0, 32), (32, 64), (64, 96), (96, 160)))
        # The gather of step i runs on the collective's own stream WHILE step i + 1 verifies
        # (elliptic_amd.sharding.OverlappedGather: double-buffered masks, all_gather_into_tensor
        # with async_op): the only exchange of the path is off the compute stream.  Every gather
        # has completed before the timed region ends (drain() + synchronize).
        from elliptic_amd.sharding import OverlappedGather
        og = OverlappedGather(nb, dist, dev, gather_device=dev if args.dist_backend == "nccl" else torch.device("cpu"))

        def strong_step():
            lane, _ = on_lane()
            with lane:
                out = og.begin()
                ctx.ecdsa_verify_dev("secp256k1", sh_h, sh_r, sh_s, sh_q, out)
                og.submit()

        drain, gathered = og.drain, og.result

        torch.cuda.synchronize()                        # the shard's copies were made on the default stream; the steps run on the lanes
        for _ in range(max(args.warmup, 2)):
            strong_step()
        drain()
        torch.cuda.synchronize()
        mask_ok = all(bool(torch.equal(gathered(b).to(exp_all.device), exp_all)) for b in (0, 1))
        if not mask_ok:                                # say where, before every rank exits on the reduced flag
            for b in (0, 1):
                bad = torch.nonzero(gathered(b).to(exp_all.device) != exp_all).flatten()
                if bad.numel():
                    sys.stderr.write("rank %d: gathered mask of buffer %d differs from the expected mask at %d of %d tuples "
                                     "(first %s; shards of %d)\n" % (rank, b, bad.numel(), nb, bad[:8].tolist(), hi - lo))
        flag = torch.tensor([1 if mask_ok else 0], dtype=torch.int64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            raise SystemExit("PARITY FAILURE: strong-scaling pass: a rank's gathered mask differs from the expected mask")
        ctx.set_timing(True)
        sdt = timed(strong_step, args.steps, drain)
        stiming = ctx.get_timing()
        ctx.set_timing(False)
        scnt, smain = stiming.get("ecdsa_main", (0, 0.0))
        # (a shard of at most three waves per SIMD runs prep and table building as ONE launch)
        spc, sprep = stiming.get("ecdsa_prep_table", stiming.get("ecdsa_prep", (0, 0.0)))
        strong_block = {"value": nb * args.steps / sdt, "unit": "verifies/s", "ms_per_step": sdt / args.steps * 1e3,
                        "global_batch": nb, "shard_rank0": hi - lo, "steps": args.steps, "warmup": args.warmup,
                        "passes_in_flight": flight,
                        "rank0_kernel_ms": {"ecdsa_main": smain / max(scnt, 1), "ecdsa_prep(+table)": sprep / max(spc, 1)},
                        "parity": "the gathered mask equals the global batch's expected mask on every rank",
                        "gather": "all_gather_into_tensor(async_op=True) on the collective's stream, double-buffered: "
                                  "step i's gather overlaps step i + 1's kernels; all gathers complete inside the timed region",
                        "workload": "BASELINE configs[2] as written: ONE batch of %d tuples sharded over %d GPUs, "
                                    "one all_gather of the masks per step" % (nb, world)}

    if rank == 0:
        total = n_global * args.steps
        value = total / dt
        cnt, main_ms = timing.get("ecdsa_main", (0, 0.0))
        pcnt, prep_ms = timing.get("ecdsa_prep", (0, 0.0))
        k_ms = main_ms / max(cnt, 1)
        # integer-VALU peak: dependency-free v_mad_u64_u32 stream on every CU
        # (best of five short runs: a single one moves by +-3 % with the clock state)
        peak_gmads = 0.0
        for _ in range(5):
            ms, ops = ctx.probe_valu(0, 256 * 8 * 4, 4096)
            peak_gmads = max(peak_gmads, ops / (ms * 1e-3) / 1e9)
        clock_ghz = torch.cuda.get_device_properties(local_rank).clock_rate / 1e6 if hasattr(
            torch.cuda.get_device_properties(local_rank), "clock_rate") else 2.4
        counters, csrc, counters_note = None, None, None
        if world == 1 and not args.no_live_counters:
            try:
                counters, csrc, counters_note = live_counters(args.batch, keep=args.keep_counters)
            except Exception as e:          # never lose the bench line to the profiler
                counters_note = "live PMC passes failed: %s" % str(e)[-200:]
        if counters is None:
            counters, csrc = kernel_counters()
        # the kernel's roofline is priced with the kernel's OWN duration: with two passes in flight the
        # events of the timed region also span the time a kernel shares the device with the other
        # pass's kernels (ecdsa_prep of step i + 1 sits behind step i's ladder for most of its length)
        k_timed, prep_timed = k_ms, prep_ms / max(pcnt, 1)
        if timing_alone is not None:
            ac, am = timing_alone.get("ecdsa_main", (0, 0.0))
            pc2, pm2 = timing_alone.get("ecdsa_prep", (0, 0.0))
            k_ms, prep_alone = am / max(ac, 1), pm2 / max(pc2, 1)
        else:
            prep_alone = prep_timed
        roof = roofline_block("ecdsa_main<secp256k1>", "verify", n, k_ms, peak_gmads, clock_ghz, counters, csrc)
        roof["prep_kernel_ms"] = prep_alone
        if timing_alone is not None:
            roof["kernel_ms_measured"] = ("HIP events around the kernel's launches in a one-stream loop of %d passes right "
                                          "after the timed region (the kernel by itself)" % alone_steps)
            roof["timed_region"] = {"passes_in_flight": flight, "kernel_ms_between_events": k_timed,
                                    "prep_kernel_ms_between_events": prep_timed,
                                    "note": "HIP events around the same kernels inside the timed region: two passes "
                                            "overlap, so these spans include shared time (what rocprofv3 --stats "
                                            "shows for the default command; --in-flight 1 reproduces kernel_ms)"}
        if roof.get("mads_issued_per_unit"):
            wj = n * roof["mads_issued_per_unit"] / (dt / args.steps) / 1e9
            roof["whole_job"] = {"achieved": wj, "frac": wj / peak_gmads if peak_gmads else None,
                                 "note": "the dominant kernel's multiplies per step / ms_per_step (prep, launch gaps and "
                                         "whatever the passes in flight hide included)"}
        roof["clock_ghz"] = clock_ghz
        if counters_note:
            roof["counters_note"] = counters_note
        out = {
            "metric": "secp256k1 ECDSA verify batch throughput (1 verify = 1 double-scalar mult u1*G+u2*Q)",
            "value": value,
            "unit": "verifies/s",
            "scalar_mults_per_s": 2 * value,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "timed_region_s": dt,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": ("secp256k1 ECDSA verify (GLV variable-base + fixed-base comb), "
                                    "batch=2^20 tuples per GPU resident in HBM, 1% corrupted"
                                    if args.batch == 1 << 20 and not strong else
                                    "secp256k1 ECDSA verify, ONE global batch of %d tuples sharded over %d GPU(s) "
                                    "(BASELINE configs[2]), resident in HBM, 1%% corrupted" % (args.batch, world)
                                    if strong else "secp256k1 ECDSA verify, batch=%d per GPU" % n),
                       "batch_per_gpu": n, "global_batch": n_global, "parallelism": "shard%d" % world,
                       "parity": "ok-mask == expected mask on all %d tuples; expected mask == oracle on the "
                                 "first %d (%s)" % (n, checked["tuples"], checked["by"]),
                       "corrupted": "every 100th tuple: one bit of z / r / s in turn, every fourth of them the key Q "
                                    "(x + 1: off the curve, %d tuples, status 2 checked against a host-side curve "
                                    "test before timing; or the next tuple's key)" % off_curve_tuples,
                       "passes_in_flight": flight,
                       "comb_bits": ctx.comb_bits("secp256k1"),
                       "library_digest": lib_digest()},
            "roofline": roof,
        }
        if one_in_flight is not None:
            out["one_in_flight"] = one_in_flight
        if sustained is not None:
            vb = ((counters or {}).get("kernels", {}).get("ecdsa_main<secp256k1>", {}) or {}).get("valu_busy") or {}
            sustained["clock_ghz_effective"] = vb.get("clock_ghz_effective")
            sustained["clock_note"] = ("GRBM_GUI_ACTIVE / 8 XCDs / the dominant kernel's wall time, both from the live PMC pass of "
                                       "this run (profiled passes clock a few per cent below un-profiled ones: MI355X_MICROARCH.md)"
                                       if vb.get("clock_ghz_effective") else "no live PMC pass in this run")
            out["sustained"] = sustained
        if per_rank is not None:
            out["per_rank"] = per_rank
        if rccl is not None:
            out["rccl"] = rccl
        if strong_block is not None:
            out["strong"] = strong_block
        if rccl_selftest is not None:
            out["rccl_selftest"] = rccl_selftest
        if world == 1 and not args.no_configs:
            out["configs"] = run_configs(ctx, dev, peak_gmads, clock_ghz, counters, csrc, h, r, s, pub, expect)
        if world == 1 and not args.no_cpu:
            base = None
            try:
                base = reference_js_baseline(h, r, s, pub, expect)
            except SystemExit:
                raise
            except Exception as e:
                base = None
                out["cpu_baseline_reference_error"] = str(e)[-300:]
            port = port_baseline(h, r, s, pub, expect)
            if base is not None:
                base["port"] = port
                out["cpu_baseline"] = base
            else:
                out["cpu_baseline"] = port
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
