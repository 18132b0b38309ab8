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
    return h.hexdigest()


def build(force=False):
    os.makedirs(OUT, exist_ok=True)
    stamp = os.path.join(OUT, "stamp")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fno-strict-aliasing", "-pthread", "-DELL_COMB_BITS_256=8",
           "-o", LIB, os.path.join(HERE, "hostsim.cpp")]
    subprocess.run(cmd, check=True)
    with open(stamp, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
