"""Build elliptic_amd/lib/libellgpu.so with hipcc for gfx950.

The kernels are instantiated per (curve, operation group) in separate
translation units (csrc/inst.hip with -DELL_INST_CURVE / -DELL_INST_GROUP) so
that they compile in parallel; objects are cached under csrc/_obj keyed by a
hash of every source file and the flags.  hipcc cross-compiles without a GPU.

    python -m elliptic_amd.build [-j N] [--force]
"""
import argparse
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libellgpu.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
FLAGS += os.environ.get("ELLGPU_CXXFLAGS", "").split()      # developer experiments (-DELL_...=...)

CURVES = ["CvP521", "CvP384", "CvP256", "CvSecp256k1", "CvP224", "CvP192"]   # slowest first
GROUPS = [4, 2, 3, 0, 1, 5, 6]
# per-group extra compiler flags: group 6 = the scalar-field kernels (see Engine::launch_fn)
GROUP_FLAGS = {6: ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def units():
    u = []
    for c in CURVES:
        for g in GROUPS:
            u.append(("inst_%s_g%d" % (c, g), "inst.hip",
                      ["-DELL_INST_CURVE=" + c, "-DELL_INST_GROUP=%d" % g] + GROUP_FLAGS.get(g, [])))
    for g in (7, 8, 10, 11, 12, 13, 14, 15, 16):
        u.append(("inst_g%d" % g, "inst.hip", ["-DELL_INST_GROUP=%d" % g]))
    u.append(("capi", "capi.hip", []))
    return u


def digest_define(name, digest):
    """capi.hip carries the source digest into the binary."""
    return ['-DELLGPU_SOURCE_DIGEST="%s"' % digest] if name.startswith("capi") else []


def source_files():
    """The explicit list of files the library is built from, as paths RELATIVE to the repo root
    (the digest must not depend on where the tree is checked out)."""
    root = os.path.normpath(os.path.join(HERE, ".."))
    files = sorted(os.path.join("elliptic_amd", "csrc", f) for f in os.listdir(CSRC)
                   if f.endswith((".h", ".hip")) and os.path.isfile(os.path.join(CSRC, f)))
    files.append(os.path.join("include", "ellgpu.h"))
    return root, files


def source_digest():
    """sha256 over (relative name, bytes) of every source file + the compiler flags.  The same
    string is compiled into the library (-DELLGPU_SOURCE_DIGEST on capi.hip) and exported as
    ellgpu_source_digest(): provenance is read from the loaded binary, not from a sidecar file."""
    h = hashlib.sha256()
    root, files = source_files()
    for f in files:
        with open(os.path.join(root, f), "rb") as fh:
            h.update(f.replace(os.sep, "/").encode() + b"\0" + fh.read() + b"\0")
    h.update((" ".join(FLAGS) + repr(sorted(GROUP_FLAGS.items()))).encode())
    return h.hexdigest()[:16]


def library_digest(path=None):
    """The digest embedded in a built libellgpu.so (None if the file is missing or predates
    ellgpu_source_digest).  Read in a CHILD process: dlopen()ing the library here would leave it
    mapped in the building process, and since glibc matches already-loaded objects by path, a
    later load of a relinked file at the same path (build() followed by smoke() in one process)
    would silently get the OLD binary back."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    code = ("import ctypes, sys\n"
            "try:\n"
            "    fn = ctypes.CDLL(sys.argv[1]).ellgpu_source_digest\n"
            "except (OSError, AttributeError):\n"
            "    sys.exit(3)\n"
            "fn.restype = ctypes.c_char_p\n"
            "sys.stdout.write(fn().decode())\n")
    try:
        p = subprocess.run([sys.executable, "-c", code, os.path.abspath(path)], capture_output=True, text=True,
                           timeout=120)
    except (OSError, subprocess.TimeoutExpired):
        return None
    out = p.stdout.strip()
    return out if p.returncode == 0 and out else None


def kernel_count(path=None):
    """distinct kernels in a built library: the kernel descriptors (`<mangled name>.kd`) of the code
    objects embedded in it"""
    import re
    path = path or LIB
    with open(path, "rb") as fh:
        return len(set(re.findall(rb"(_ZN3ell\w+)\.kd\0", fh.read())))


STATS = os.path.join(LIBDIR, "build_stats.json")


def build_stats():
    """what the last build of the library cost (written by build(): lib/build_stats.json, beside the
    git-ignored .so): cold / warm seconds, translation units, kernels, bytes.  None before a build."""
    import json
    try:
        with open(STATS) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def compile_one(args):
    name, src, defs, digest, extra = args
    obj = os.path.join(OBJ, "%s.%s.o" % (name, digest))
    if os.path.exists(obj):
        return name, obj, 0.0, ""
    t0 = time.time()
    cmd = [HIPCC] + FLAGS + extra + defs + digest_define(name, digest) + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (name, " ".join(cmd), p.stderr[-4000:]))
    os.replace(obj + ".tmp", obj)
    return name, obj, time.time() - t0, p.stderr


def build_dev_k256(verbose=True, curve="CvSecp256k1"):
    """secp256k1-only developer library (lib/libellgpu_dev.so, ~40 s): use with
    ELLGPU_LIB=elliptic_amd/lib/libellgpu_dev.so for kernel iteration."""
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    digest = "dev" + source_digest()
    cid = {"CvSecp256k1": 0, "CvP192": 1, "CvP224": 2, "CvP256": 3, "CvP384": 4, "CvP521": 5}[curve]
    dflag = "-DELL_ONLY_CURVE=%d" % cid
    tflag = "-DELL_ONLY_TYPE=" + curve
    work = [("inst_%s_g%d" % (curve, g), "inst.hip", ["-DELL_INST_CURVE=" + curve, "-DELL_INST_GROUP=%d" % g,
                                                      dflag, tflag] + GROUP_FLAGS.get(g, []), digest,
             ["-Rpass-analysis=kernel-resource-usage"]) for g in GROUPS]
    for g in (7, 8, 10, 11, 12, 13, 14, 15, 16):   # ed25519 / x25519 units are referenced by the engine, keep them linkable
        work.append(("inst_g%d" % g, "inst.hip", ["-DELL_INST_GROUP=%d" % g, dflag, tflag], digest, []))
    work.append(("capi_%s" % curve, "capi.hip", [dflag, tflag], digest, []))
    for f in os.listdir(OBJ):
        if ".dev" in f and digest not in f:
            os.remove(os.path.join(OBJ, f))
    t0 = time.time()
    with cf.ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(compile_one, work))
    objs = [r[1] for r in res]
    with open(os.path.join(OBJ, "resource_usage_dev.log"), "w") as f:
        f.write("\n".join(r[3] for r in res))
    out = os.path.join(LIBDIR, "libellgpu_dev.so")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs,
                       capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("link failed:\n" + p.stderr[-4000:])
    if verbose:
        print("built %s in %.1fs" % (out, time.time() - t0))
    return out


def build(jobs=None, force=False, verbose=True, remarks=False):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    digest = source_digest()
    t_start = time.time()
    if not force and library_digest(LIB) == digest:
        if verbose:
            st = build_stats() or {}
            print("libellgpu.so up to date (%s): checked in %.1f s; %s kernels, %.1f MB; its build took %s s "
                  "(%s translation units compiled, %d-way parallel)"
                  % (digest, time.time() - t_start, kernel_count(), os.path.getsize(LIB) / 1e6,
                     st.get("seconds", "?"), st.get("units_compiled", "?"), st.get("jobs", 0)))
        return LIB
    # drop objects of older source states
    for f in os.listdir(OBJ):
        if digest not in f and ".dev" not in f and f.endswith(".o"):
            os.remove(os.path.join(OBJ, f))
    jobs = jobs or max(1, (os.cpu_count() or 2))
    extra = ["-Rpass-analysis=kernel-resource-usage"] if remarks else []
    work = [(n, s, d, digest, extra) for (n, s, d) in units()]
    objs = []
    log = []
    t0 = time.time()
    compiled, cpu_s = 0, 0.0
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for name, obj, dt, err in ex.map(compile_one, work):
            objs.append(obj)
            log.append(err)
            if dt:
                compiled += 1
                cpu_s += dt
            if verbose and dt:
                print("  compiled %-28s %6.1fs" % (name, dt), flush=True)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode != 0:
        raise RuntimeError("link failed:\n" + p.stderr[-4000:])
    os.replace(LIB + ".tmp", LIB)
    if remarks:
        with open(os.path.join(OBJ, "resource_usage.log"), "w") as f:
            f.write("\n".join(log))
    # what the build cost, for the record (README "Build"): a cold build compiles every unit
    import json
    stats = {"digest": digest, "seconds": round(time.time() - t0, 1), "units": len(work), "units_compiled": compiled,
             "cold": compiled == len(work), "sum_of_unit_seconds": round(cpu_s, 1), "jobs": jobs,
             "host_cpus": os.cpu_count(), "kernels": kernel_count(), "library_bytes": os.path.getsize(LIB)}
    with open(STATS, "w") as fh:
        json.dump(stats, fh)
    if verbose:
        print("built %s in %.1f s (%s: %d of %d translation units compiled, %.0f s of compiler time, %d-way parallel on "
              "%d CPUs): %d kernels, %.1f MB" % (LIB, stats["seconds"], "cold" if stats["cold"] else "warm", compiled,
                                                len(work), cpu_s, jobs, os.cpu_count() or 0, stats["kernels"],
                                                stats["library_bytes"] / 1e6))
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=None)
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--remarks", action="store_true", help="keep kernel-resource-usage remarks in csrc/_obj/resource_usage.log")
    ap.add_argument("--dev-k256", action="store_true", help="secp256k1-only developer library")
    ap.add_argument("--dev", default=None, help="single-curve developer library, e.g. --dev CvP384")
    a = ap.parse_args()
    if a.dev_k256 or a.dev:
        build_dev_k256(curve=a.dev or "CvSecp256k1")
    else:
        build(a.j, a.force, True, a.remarks)
