"""CPU-side checks of the drop-in boundary: libellgpu.so (cross-compiled for
gfx950 by elliptic_amd/build.py) loads and exports every symbol include/ellgpu.h
declares; without a GPU the product path fails loudly instead of falling back."""
import ctypes
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ellgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ellgpu_\w+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib_path():
    from elliptic_amd import build as b
    if not os.path.exists(b.LIB):
        if shutil.which("hipcc") is None and not os.path.exists(b.HIPCC):
            pytest.skip("hipcc not available and libellgpu.so not built")
        b.build(verbose=False)
    return b.LIB


def test_header_symbols_all_exported(lib_path):
    from elliptic_amd import _lib
    lib = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "libellgpu.so does not export %s" % n
    # and the Python binding table covers the header
    assert set(names) == set(_lib.SYMBOLS)


def test_metadata_calls_work_without_gpu(lib_path):
    from elliptic_amd import _lib
    lib = _lib.load(lib_path)
    assert lib.ellgpu_version() == 0x000200 == _lib.ABI_VERSION
    for i, (name, fb) in enumerate([("secp256k1", 32), ("p192", 24), ("p224", 28), ("p256", 32),
                                    ("p384", 48), ("p521", 66), ("ed25519", 32), ("curve25519", 32)]):
        assert lib.ellgpu_curve_id(name.encode()) == i
        assert lib.ellgpu_curve_field_bytes(i) == fb
        assert lib.ellgpu_curve_order_bytes(i) == fb
    assert lib.ellgpu_curve_id(b"brainpool") == -1


def test_source_digest_is_embedded_and_path_independent(lib_path, tmp_path):
    """the library carries the digest of the sources it was built from (ellgpu_source_digest),
    and that digest does not depend on where the tree lives: a copy of the sources elsewhere
    hashes to the same value (round 2's smoke gate failed on exactly this)"""
    import subprocess
    import sys
    from elliptic_amd import build as b, _lib
    lib = _lib.load(lib_path)
    emb = lib.ellgpu_source_digest().decode()
    assert re.fullmatch(r"[0-9a-f]{16}", emb), emb
    assert emb == b.source_digest() == b.library_digest(lib_path)
    root, files = b.source_files()
    assert all(not os.path.isabs(f) for f in files) and files == sorted(files[:-1]) + [files[-1]]
    # same sources at another path (and reached through a symlink) -> same digest
    dst = tmp_path / "elsewhere" / "copy"
    for f in files + [os.path.join("elliptic_amd", "build.py"), os.path.join("elliptic_amd", "__init__.py")]:
        os.makedirs(os.path.dirname(dst / f), exist_ok=True)
        shutil.copyfile(os.path.join(root, f), dst / f)
    open(dst / "elliptic_amd" / "__init__.py", "w").close()
    os.symlink(dst, tmp_path / "link")
    for where in (dst, tmp_path / "link"):
        out = subprocess.run([sys.executable, "-c", "from elliptic_amd import build as b; print(b.source_digest())"],
                             cwd=str(where), capture_output=True, text=True, check=True,
                             env={**os.environ, "PYTHONPATH": str(where)})
        assert out.stdout.strip() == emb


def test_no_cpu_fallback(lib_path):
    """on a box without an MI355X the product must refuse to run"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import elliptic_amd
    with pytest.raises(elliptic_amd.EllgpuError) as e:
        elliptic_amd.Context(0)
    assert e.value.code == -1


def test_product_does_not_reference_oracle():
    """nothing under elliptic_amd/ or include/ may import or link oracle/"""
    bad = []
    for base in ("elliptic_amd", "include"):
        for dp, dn, fn in os.walk(os.path.join(ROOT, base)):
            if "_obj" in dp or "node_modules" in dp:
                continue
            for f in fn:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c", ".js")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"(from|import)\s+oracle|oracle/|ec_oracle|hostsim\.so|libellgpu_hostsim", txt):
                        if f == "common.h" or f == "engine.h" or f == "capi_common.h":
                            # comments that NAME the test build are fine; code references are not
                            code = re.sub(r"//.*", "", txt)
                            if not re.search(r"oracle|hostsim", code):
                                continue
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_reading_the_digest_does_not_map_the_library(tmp_path):
    """build.library_digest() must not dlopen the library in the calling process: glibc matches
    loaded objects by path, so a relinked file at the same path would later resolve to the OLD
    mapping (build() then smoke() in one process).  Done with two toy libraries in a child process:
    read the digest of the first, replace the file, load it -- the NEW code must answer."""
    import subprocess
    import sys
    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    so = tmp_path / "libtoy.so"
    for tag in ("old", "new"):
        c = tmp_path / ("toy_%s.c" % tag)
        c.write_text('const char* ellgpu_source_digest(void) { return "%s"; }\n' % tag)
        subprocess.run(["gcc", "-shared", "-fPIC", "-o", str(tmp_path / ("libtoy_%s.so" % tag)), str(c)], check=True)
    code = (
        "import ctypes, os, shutil, sys\n"
        "sys.path.insert(0, %r)\n"
        "from elliptic_amd import build as b\n"
        "d, so = sys.argv[1], sys.argv[2]\n"
        "shutil.copy(os.path.join(d, 'libtoy_old.so'), so)\n"
        "assert b.library_digest(so) == 'old'\n"
        "assert not any('libtoy' in l for l in open('/proc/self/maps'))\n"
        "shutil.copy(os.path.join(d, 'libtoy_new.so'), so + '.tmp'); os.replace(so + '.tmp', so)\n"
        "fn = ctypes.CDLL(so).ellgpu_source_digest; fn.restype = ctypes.c_char_p\n"
        "assert fn() == b'new', fn()\n" % ROOT)
    p = subprocess.run([sys.executable, "-c", code, str(tmp_path), str(so)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
