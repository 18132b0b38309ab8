"""oracle/_ref: the reference itself, placed next to the checker.  TEST INFRASTRUCTURE ONLY.

The reference (indutny/elliptic 6.6.1) is JavaScript: there is nothing to compile, but the
GPU box has no /root/reference, so its self-contained browserify bundle
(dist/elliptic.js = lib/ + vendored bn.js 4.11.9, hash.js, hmac-drbg, brorand)
and its own mocha spec files + fixtures (test/) are copied -- by THIS recipe, from where they
lie under /root/reference -- into the git-ignored oracle/_ref/, which travels to the GPU box
like the built .so files do.  Nothing of it enters the repository's history and the product
(elliptic_amd/) never reads it; users are

  * tests/test_js_install.py (-m gpu): the reference's unmodified suite and the golden replay
    run through install() on the real libellgpu.so, with ELLIPTIC_REFERENCE=oracle/_ref;
  * bench.py cpu_baseline (kind "reference"): ec.verify of the reference under Node on the GPU
    box's host cores, on a slice of the very tuples the GPU verifies.

    python -m oracle.make_ref          (also called by __graft_entry__.build())
"""
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = os.environ.get("ELLIPTIC_REFERENCE_SRC", "/root/reference")
REF_DIR = os.path.join(HERE, "_ref")
FILES = ["dist/elliptic.js", "package.json"]
DIRS = ["test"]


def present():
    """the copied reference, or None (callers skip / fall back to the C port)"""
    return REF_DIR if os.path.exists(os.path.join(REF_DIR, "dist", "elliptic.js")) else None


def build(verbose=False):
    """copy the bundle + spec files when the reference checkout is here; returns the
    directory, or whatever an earlier run left (None if nothing)"""
    if not os.path.exists(os.path.join(REF_SRC, "dist", "elliptic.js")):
        return present()
    for f in FILES:
        dst = os.path.join(REF_DIR, f)
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        src = os.path.join(REF_SRC, f)
        if not os.path.exists(dst) or os.path.getmtime(dst) < os.path.getmtime(src):
            shutil.copy2(src, dst)
    for d in DIRS:
        dst = os.path.join(REF_DIR, d)
        if os.path.exists(dst):
            shutil.rmtree(dst)
        shutil.copytree(os.path.join(REF_SRC, d), dst)
    if verbose:
        print("oracle/_ref: reference bundle + spec files copied from", REF_SRC)
    return REF_DIR


if __name__ == "__main__":
    print(build(verbose=True))
