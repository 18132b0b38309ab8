"""Build the N-API addon elliptic_amd/js/ellgpu.node with gcc (plain C,
node_api.h from /usr/include/node; node-gyp is not needed)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "ellgpu.node")
SRC = os.path.join(HERE, "ellgpu_napi.c")


def node_include():
    for d in ("/usr/include/node", "/usr/local/include/node"):
        if os.path.exists(os.path.join(d, "node_api.h")):
            return d
    return None


def build(force=False):
    inc = node_include()
    if inc is None or shutil.which("gcc") is None:
        return None
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    subprocess.run(["gcc", "-shared", "-fPIC", "-O2", "-I", inc, "-DNODE_GYP_MODULE_NAME=ellgpu",
                    "-o", OUT, SRC, "-ldl"], check=True)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
