"""The JavaScript host layer (elliptic_amd/js: N-API addon + install() patch).

CPU: the reference's own, unmodified mocha suite must pass with install()
applied to the reference library; in this GPU-less container the addon is
pointed at the CPU unit-test build of the device code (tests/hostsim), so what
is exercised is the marshalling, the prototype patch and the device code's
logic.  Needs Node and a copy of the reference: /root/reference (build container)
or the git-ignored oracle/_ref that oracle/make_ref.py fills from it and that travels
to the GPU box.
GPU (-m gpu): the same two gates -- the reference's unmodified suite and the golden
replay incl. every exception message -- through install() on the REAL libellgpu.so
(generated-asm multiplier, 16-bit comb: the shipped hot path), and the batch API
through the addon against the golden fixtures."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


# On the CPU build p384 / p521 single calls stay on the one-item-per-lane kernels: the host simulation of
# their 64-lane wide layer (csrc/coop_wide.h) costs ~50 ms per verify, and these replays make hundreds;
# tests/test_hostsim_golden.py runs the wide layer on the same fixtures, the -m gpu tests run it for real.
HOSTSIM_ENV = {"ELLGPU_WIDE_GRID": "0"}


def _reference():
    """directory of a reference copy (ELLIPTIC_REFERENCE for tools/ref_loader.js), or skip"""
    if os.path.exists("/root/reference/dist/elliptic.js"):
        return "/root/reference"
    from oracle import make_ref
    d = make_ref.present()
    if d is None:
        pytest.skip("no copy of the reference (neither /root/reference nor oracle/_ref)")
    return d


def _run_suite(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "run_ref_tests_patched.js")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["failed"] == 0 and res["passed"] == res["total"] >= 226
    # the patched methods really were the ones running
    # (pass-through: toy curves, and the RFC 6979 vectors that sign with another hash than the preset's)
    assert res["engine"]["gpuCalls"] > 500 and res["engine"]["passthrough"] < 100
    return res


def _run_replay(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_patched_results.js")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-3000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["checked"] > 1000 and res["thrown"] > 300
    return res


def _addon():
    from elliptic_amd.js import build as jb
    p = jb.build()
    if p is None:
        pytest.skip("node headers / gcc not available")
    return p


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_reference_suite_passes_with_install_patch():
    _addon()
    from hostsim.build import build as build_hostsim
    _run_suite(build_hostsim())


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_patched_api_matches_goldens_including_exception_messages():
    """recoverPubKey / EDDSA sign / EDDSA verify / pointFromX / pointFromY through install():
    the results and the message of every thrown Error equal what the unpatched reference
    produced when the golden files were generated"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_replay(build_hostsim())


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
@pytest.mark.parametrize("custom", ["0", "1"])
def test_custom_generator_curves_pass_through(custom):
    """a curve over a preset's field with ANOTHER generator (or none) must not be served from the
    preset's tables, which belong to the preset's G and n: with customCurves off it
    stays on the reference's code, by default it takes the device's user-defined-curve path;
    results equal the unpatched reference either way"""
    _addon()
    from hostsim.build import build as build_hostsim
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference(), ELLGPU_LIB=build_hostsim(), ELLGPU_CUSTOM=custom, **HOSTSIM_ENV)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_custom_generator.js")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["checked"] >= 30


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_reference_suite_passes_with_install_patch_gpu():
    """the reference's own 226 specs (RFC 6979, Maxwell-trick, Wycheproof, sign.input ...) with
    install() routing the hot path into the real libellgpu.so on the MI355X"""
    _addon()
    _run_suite(None)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_patched_api_matches_goldens_including_exception_messages_gpu():
    _addon()
    _run_replay(None)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_user_defined_curves_through_install_gpu():
    """custom-generator / generator-less curves through install() on the real library: the
    user-defined-curve kernels (run-time prime, generic-a doubling) against the unpatched reference"""
    _addon()
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_custom_generator.js")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["checked"] >= 30 and out["custom"] is True


def _run_coalescing(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_verify_coalescing.js")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["calls"] == 103 and res["engine_calls"] <= 8 and res["rejected"] == 2
    return res


def _run_fuzz(lib, iterations, seed):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "fuzz_patched_vs_plain.js"), str(iterations), seed], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["calls"] >= iterations and res["reference_threw"] > iterations // 10
    assert res["engine"]["offCurve"] > 0 and res["engine"]["passthrough"] > 0


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_differential_fuzz_of_the_public_api():
    """the same seeded sequence of public-API calls (mul / mulAdd / jmulAdd, ECDSA sign / verify /
    recoverPubKey / derive, pointFromX / pointFromY, EdDSA sign / verify) with arguments on the seams
    (edge, negative and over-wide scalars, infinity, tabled, off-curve and non-canonical points,
    messages and signatures in every accepted form, corrupted ones) on an unpatched and a patched
    copy of the reference: every result and every exception message equal"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_fuzz(build_hostsim(), 200, "ci-1")


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_differential_fuzz_of_the_public_api_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_fuzz(None, 600, "gpu-1")


def _run_close(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_engine_close.js")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["refused_while_in_flight"] and res["engines_opened_and_closed"] == 70


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_engine_close_releases_the_pinned_context():
    """the addon pins a context until destroyContext (node 12 frees an environment with a pending
    weak callback of a collected external otherwise: INTEGRATION.md section 2): Engine#close releases it,
    is refused while a Promise-form batch is in flight, leaves the library the reference's after
    uninstall(), and more engines than the addon pins at once can be opened one after the other"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_close(build_hostsim())


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_engine_close_releases_the_pinned_context_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_close(None)


def _run_walk(lib, families, stride=1, offset=0, timeout=3000):
    """tools/probe_mutation_walk.js: ONE reachable property of the caller's objects changed after their
    first use -- every property the walk finds from an EC / EDDSA instance, its curve, the generator's and
    an operand's tables, a key pair, a signature -- calls, undo, calls: the patched library answers what the
    reference answers, except inside what INTEGRATION.md section 2 lists as treated as immutable"""
    # (a change that makes the REFERENCE loop costs two time limits: a path takes well under a second)
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference(), WALK_STRIDE=str(stride), WALK_OFFSET=str(offset),
               WALK_LIMIT_MS=os.environ.get("WALK_LIMIT_MS", "15000" if lib is None else "9000"))
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    procs = [subprocess.Popen(["node", os.path.join(ROOT, "tools", "probe_mutation_walk.js"), f], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for f in families]
    rows = []
    for f, p in zip(families, procs):
        try:
            out, err = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            raise
        assert p.returncode == 0, f + "\n" + out[-3000:] + err[-2000:]
        res = json.loads(out.strip().splitlines()[-1])
        row = json.loads([l for l in out.splitlines() if l.startswith('{"family"')][0])
        assert res["failures"] == 0 and row["failures"] == 0 and row["same"] > 0, out[-3000:]
        # most changes DO change the reference's answers: the walk is not comparing two constants
        assert row["paths"] < 30 or row["same_and_the_reference_answer_changed"] * 4 > row["same"], row
        rows.append(row)
    return rows


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_mutation_walk_over_everything_reachable():
    """every eleventh path of the secp256k1, ed25519 and curve25519 walks on the CPU build of the device
    code (the GPU suite runs all of them, p256 too)"""
    _addon()
    from hostsim.build import build as build_hostsim
    rows = _run_walk(build_hostsim(), ["short:secp256k1", "edwards:ed25519", "mont:curve25519"], stride=11, offset=3)
    assert rows[0]["paths"] >= 39 and rows[1]["paths"] >= 54 and rows[2]["paths"] >= 7, rows


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_mutation_walk_over_everything_reachable_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    rows = _run_walk(None, ["short:secp256k1", "short:p256", "edwards:ed25519", "mont:curve25519"])
    assert sum(r["paths"] for r in rows) >= 1400, rows


def _run_eddsa_edges(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    p = subprocess.run(["node", os.path.join(ROOT, "tools", "check_eddsa_edge_encodings.js")], env=env,
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    res = json.loads(p.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["cases"] >= 2000 and res["accepted"] >= 10 and res["thrown"] >= 100


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_eddsa_verify_edge_encodings():
    """EDDSA#verify on low-order and non-canonical encodings of A and R (y = 0, 1, p, p + 1, the
    order-8 points ...), both sign bits, S in {0, 1, 5, n - 1, n}: patched library and batch API
    against the unpatched reference, verdicts and throws"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_eddsa_edges(build_hostsim())


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_eddsa_verify_edge_encodings_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_eddsa_edges(None)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_verify_async_coalesces_concurrent_calls():
    """99 concurrent eng.verifyAsync calls (two curves, two digest lengths, corrupted tuples, two
    calls the reference throws on) -> at most four engine calls, every verdict and every rejection
    message equal to the unpatched reference's synchronous EC#verify"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_coalescing(build_hostsim())


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_verify_async_coalesces_concurrent_calls_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_coalescing(None)


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_batch_api_hostsim():
    _addon()
    from hostsim.build import build as build_hostsim
    p = subprocess.run(["node", os.path.join(ROOT, "elliptic_amd", "js", "selftest.js"), build_hostsim()],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["checked"] > 1000


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_js_batch_api_gpu():
    _addon()
    p = subprocess.run(["node", os.path.join(ROOT, "elliptic_amd", "js", "selftest.js")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert json.loads(p.stdout.strip().splitlines()[-1])["checked"] > 1000


def _run_toy_probe(lib, p, kind, min_calls):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    q = subprocess.run(["node", os.path.join(ROOT, "tools", "probe_toy_curves.js"), str(p), kind], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
    res = json.loads(q.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["mismatches"] == 0 and res["calls"] >= min_calls and res["engine_calls"] > 0, res


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
@pytest.mark.parametrize("p,kind,min_calls", [(5, "short", 3000), (5, "edwards", 1500), (7, "edwards", 6000)])
def test_every_toy_curve_every_point(p, kind, min_calls):
    """EVERY curve over F_p in the given model (singular cubics and Edwards curves without a complete
    addition law included), every point on it, k = 0 .. 2p + 3, mulAdd / jmulAdd grids: the patched
    library against the unpatched reference (tools/probe_toy_curves.js) -- the curves whose equation
    gives no (complete) group law must stay on the reference's own code"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_toy_probe(build_hostsim(), p, kind, min_calls)


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_every_toy_curve_every_point_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_toy_probe(None, 7, "short", 12000)
    _run_toy_probe(None, 7, "edwards", 6000)


def _run_api_walk(lib):
    env = dict(os.environ, ELLIPTIC_REFERENCE=_reference())
    if lib:
        env["ELLGPU_LIB"] = lib
        env.update(HOSTSIM_ENV)
    else:
        env.pop("ELLGPU_LIB", None)
    q = subprocess.run(["node", os.path.join(ROOT, "tools", "probe_api_walk.js")], env=env,
                       capture_output=True, text=True, timeout=1500)
    assert q.returncode == 0, q.stdout[-2000:] + q.stderr[-2000:]
    res = json.loads(q.stdout.strip().splitlines()[-1])
    assert res["ok"] and res["calls"] >= 600 and res["reference_threw"] >= 150 and res["engine"]["gpuCalls"] >= 400, res


@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_public_api_walk_every_input_form():
    """One set of inputs in every form the reference accepts -- keys, signatures, messages, recovery
    ids, pointFromX / decodePoint forms on four curves, EdDSA keys / signatures / secrets / messages
    -- on an unpatched and on a patched copy (tools/probe_api_walk.js): 635 calls, every result and
    every exception message identical"""
    _addon()
    from hostsim.build import build as build_hostsim
    _run_api_walk(build_hostsim())


@pytest.mark.gpu
@pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")
def test_public_api_walk_every_input_form_gpu():
    from elliptic_amd.js import build as jb
    jb.build()
    _run_api_walk(None)
