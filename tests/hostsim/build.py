"""Build tests/hostsim/_build/libellgpu_hostsim.so with g++ (CPU-only unit-test
build of the device headers; see hostsim.cpp)."""
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libellgpu_hostsim.so")


def _digest():
    h = hashlib.sha256()
    src = [os.path.join(HERE, "hostsim.cpp"), os.path.join(ROOT, "include", "ellgpu.h")]
    cs = os.path.join(ROOT, "elliptic_amd", "csrc")
    src += sorted(os.path.join(cs, f) for f in os.listdir(cs) if f.endswith(".h"))
    for f in src:
        with open(f, "rb") as fh:
            h.update(fh.read())
    h.update(b"flags:-DELL_COMB_BITS_256=8 -DELL_BOUNDS_CHECK=1 -DELL_COMB_SLICE=1000")
    return h.hexdigest()


def build(force=False, lazy_k256=False):
    """lazy_k256: the variant in which secp256k1 runs on the 9 x 29-bit field (-DELL_K256_LAZY=1,
    fpk256l.h) -- a build switch of the product that the default build does not exercise"""
    os.makedirs(OUT, exist_ok=True)
    tag = "_lazy" if lazy_k256 else ""
    lib = LIB.replace(".so", tag + ".so")
    stamp = os.path.join(OUT, "stamp" + tag)
    dig = _digest()
    if not force and os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == dig:
        return lib
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-strict-aliasing", "-pthread", "-DELL_COMB_BITS_256=8", "-DELL_BOUNDS_CHECK=1",
           "-DELL_COMB_SLICE=1000"]          # narrow combs built in several slices (ensure_comb's `first` offsets)
    if lazy_k256:
        cmd.append("-DELL_K256_LAZY=1")
    cmd += ["-o", lib, os.path.join(HERE, "hostsim.cpp")]
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return lib


if __name__ == "__main__":
    print(build(force=True))
