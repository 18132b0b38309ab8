#!/bin/sh
# AddressSanitizer + UBSan run of the host-side engine logic and the device headers'
# CPU unit-test build (tests/hostsim) over golden vectors of several curves.
# Build container only (needs g++ with libasan).   sh tools/asan_check.sh
set -e
cd "$(dirname "$0")/.."
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -fno-omit-frame-pointer -fsanitize=address,undefined \
    -DELL_COMB_BITS_256=8 -DELL_COMB_SLICE=1000 -o /tmp/ellgpu_hs_asan.so tests/hostsim/hostsim.cpp
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" python - <<'PY'
import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import elliptic_amd
from elliptic_amd import _lib
import parity_checks as PC
lib = _lib.load('/tmp/ellgpu_hs_asan.so', optional=("ellgpu_probe_valu", "ellgpu_ctx_set_timing",
                                                    "ellgpu_ctx_get_timing", "ellgpu_debug_field_op"))
ctx = elliptic_amd.Context(0, lib_path=lib)
for c in ["secp256k1", "p224", "p521", "ed25519"]:
    print(c, "mul", PC.check_mul_golden(ctx, c))
print("x25519", PC.check_x25519_golden(ctx))
for c in ["secp256k1", "p384", "p521"]:
    print(c, "verify", PC.check_verify_golden(ctx, c))
for c in ["secp256k1", "p256", "p521"]:
    print(c, "sign", PC.check_sign_golden(ctx, c), "recover", PC.check_recover_golden(ctx, c),
          "decompress", PC.check_decompress_golden(ctx, c))
for c in ["secp256k1", "p224", "p521", "ed25519"]:
    print(c, "codec", PC.check_codec_golden(ctx, c))
for c in ["secp256k1", "p224", "p521"]:
    print(c, "wire", PC.check_wire_golden(ctx, c))
print("eddsa verify", PC.check_eddsa_golden(ctx), "sign", PC.check_eddsa_sign_golden(ctx))
# round 4: the off-curve domain marks (all entry points), the two-kernel small-grid verify, the
# pipelined normalization and the directed exceptional keys
for c in ["secp256k1", "p224", "p384", "ed25519"]:
    print(c, "off-curve", PC.check_offcurve_golden(ctx, c))
for c in ["secp256k1", "p256"]:
    print(c, "exceptional keys", PC.check_exceptional_keys(ctx, c))
for spec in PC.custom_curves()[:2]:
    print(spec["name"], "custom", PC.check_custom_short_golden(ctx, spec))
for spec in PC.custom_edwards_curves()[:1]:
    print(spec["name"], "custom edwards", PC.check_custom_edwards_golden(ctx, spec))
ctx.close()
print("ASAN/UBSAN run clean")
PY
