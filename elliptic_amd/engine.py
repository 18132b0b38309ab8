"""Host-side batch engine over the C ABI (include/ellgpu.h).

One `Context` = one MI355X + one HIP stream (one process per GPU).  Two calling
styles, matching the two halves of the C ABI:

  * host buffers (numpy uint8 arrays / bytes): `mul_fixed`, `mul_var`,
    `mul_add2`, `ecdsa_verify`, `x25519` -- synchronous, what the N-API addon
    does for the patched JS library;
  * device buffers (torch.uint8 CUDA tensors): the `*_dev` methods -- inputs and
    outputs stay resident in HBM, work is enqueued on torch's current stream
    (bench.py, multi-GPU sharding).

All integers are fixed-width big-endian byte strings (see include/ellgpu.h).
"""
import ctypes

import numpy as np

from . import _lib

CURVES = ["secp256k1", "p192", "p224", "p256", "p384", "p521", "ed25519", "curve25519"]
CURVE_ID = {n: i for i, n in enumerate(CURVES)}
FIELD_BYTES = {"secp256k1": 32, "p192": 24, "p224": 28, "p256": 32, "p384": 48, "p521": 66,
               "ed25519": 32, "curve25519": 32}
ORDER_BYTES = dict(FIELD_BYTES)
# per-item status in the `inf` / `ok` results (include/ellgpu.h ELLGPU_STATUS_OFF_CURVE): a point
# operand is not on the curve -- outside the engine's domain, reported instead of guessed (the
# reference computes with such points, and its answer depends on the order of its own operations)
STATUS_OFF_CURVE = 2
# user-defined short curves (Context.define_short) are addressed by the integer id the library
# hands out; their scalars and coordinates are 32 bytes wide whatever the prime's size
CURVE_CUSTOM0 = 16
for _i in range(CURVE_CUSTOM0, CURVE_CUSTOM0 + 16):
    FIELD_BYTES[_i] = ORDER_BYTES[_i] = 32


def _u8(a, shape=None):
    if isinstance(a, (bytes, bytearray, memoryview)):
        a = np.frombuffer(bytes(a), dtype=np.uint8)
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if shape is not None:
        a = a.reshape(shape)
    return a


def ints_to_be(values, width):
    """list of ints -> (n, width) uint8 big-endian array"""
    out = np.zeros((len(values), width), dtype=np.uint8)
    for i, v in enumerate(values):
        out[i] = np.frombuffer(int(v).to_bytes(width, "big"), dtype=np.uint8)
    return out


def be_to_ints(arr):
    arr = np.asarray(arr, dtype=np.uint8)
    return [int.from_bytes(row.tobytes(), "big") for row in arr]


class Context:
    def __init__(self, device=0, lib_path=None, devices=None):
        """device: one HIP ordinal; devices=[...]: a GROUP over several (ellgpu_group_create) --
        mul_fixed / mul_var / mul_add2 / ecdsa_verify on host buffers are then sharded over the
        devices, one host thread each, results copied straight into the caller's arrays"""
        self._lib = _lib.load(lib_path) if not hasattr(lib_path, "ellgpu_version") else lib_path
        self._ctx = ctypes.c_void_p()
        self._own = None
        self._pending_default = None
        if devices is not None:
            devs = [int(d) for d in devices]
            arr = (ctypes.c_int * len(devs))(*devs)
            rc = self._lib.ellgpu_group_create(arr, len(devs), ctypes.byref(self._ctx))
            device = devs[0] if devs else 0
        else:
            rc = self._lib.ellgpu_ctx_create(int(device), ctypes.byref(self._ctx))
        if rc != 0:
            raise _lib.EllgpuError(rc, self._lib.ellgpu_last_error().decode())
        self.device = device
        self.devices = list(devices) if devices is not None else [device]
        self._own = None                  # torch view of the context's own stream (see _stream)
        self._pending_default = None

    def group_size(self):
        return self._lib.ellgpu_group_size(self._ctx)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.ellgpu_ctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if self._pending_default is not None:              # see _stream()
            cur, self._pending_default = self._pending_default, None
            cur.wait_stream(self._own)
        if rc != 0:
            raise _lib.EllgpuError(rc, self._lib.ellgpu_last_error().decode())

    @staticmethod
    def _cid(curve):
        return CURVE_ID[curve] if isinstance(curve, str) else int(curve)

    def synchronize(self):
        self._check(self._lib.ellgpu_ctx_synchronize(self._ctx))

    def define_short(self, p, a, b):
        """Register y^2 = x^3 + a x + b over the odd prime p < 2^256 (`new curve.short({p, a, b})`
        with parameters that are no preset, lib/elliptic/curve/short.js:11-24) and return its
        curve id: valid for mul_var / mul_add2 (both points given) / point_add and their _dev forms."""
        cid = ctypes.c_int(-1)
        enc = [int(v % p if i else v).to_bytes(32, "big") for i, v in enumerate((int(p), int(a), int(b)))]
        self._check(self._lib.ellgpu_curve_define_short(self._ctx, enc[0], enc[1], enc[2], ctypes.byref(cid)))
        return cid.value

    def define_edwards(self, p, a, d):
        """Register a x^2 + y^2 = 1 + d x^2 y^2 over the odd prime p < 2^256 (`new curve.edwards({p, a,
        c: 1, d})` with parameters that are not ed25519's) and return its curve id (as define_short)."""
        cid = ctypes.c_int(-1)
        enc = [int(v % p if i else v).to_bytes(32, "big") for i, v in enumerate((int(p), int(a), int(d)))]
        self._check(self._lib.ellgpu_curve_define_edwards(self._ctx, enc[0], enc[1], enc[2], ctypes.byref(cid)))
        return cid.value

    def reserve(self, curve, n):
        self._check(self._lib.ellgpu_ctx_reserve(self._ctx, self._cid(curve), int(n)))

    def comb_bits(self, curve):
        """window width of the curve's fixed-base table on this context (0 = not built yet; the
        256-bit curves default to 22, narrower when the device could not hold that table)"""
        rc = self._lib.ellgpu_ctx_comb_bits(self._ctx, self._cid(curve))
        if rc < 0:
            raise _lib.EllgpuError(rc, self._lib.ellgpu_last_error().decode())
        return rc

    def defer(self):
        """the next few-item host-buffer call on this context returns once its work is enqueued;
        its result arrays are filled by collect() (ellgpu_ctx_defer / ellgpu_ctx_collect: the
        split form behind install()'s table re-validation).  Keep the result arrays alive."""
        self._check(self._lib.ellgpu_ctx_defer(self._ctx))

    def collect(self):
        self._check(self._lib.ellgpu_ctx_collect(self._ctx))

    # ---- host buffers -------------------------------------------------------

    def mul_fixed(self, curve, k, out=None):
        """out: optional (xy, inf) uint8 arrays to write into (see mul_var)"""
        B = FIELD_BYTES[curve]
        k = _u8(k, (-1, B))
        n = k.shape[0]
        if out is None:
            out, inf = np.zeros((n, 2 * B), np.uint8), np.zeros(n, np.uint8)
        else:
            out, inf = out
            assert out.dtype == np.uint8 and out.shape == (n, 2 * B) and out.flags.c_contiguous
            assert inf.dtype == np.uint8 and inf.shape == (n,) and inf.flags.c_contiguous
        self._check(self._lib.ellgpu_mul_fixed(self._ctx, self._cid(curve), n, k.ctypes.data,
                                               out.ctypes.data, inf.ctypes.data))
        return out, inf

    def mul_var(self, curve, k, xy, out=None):
        """out: optional (xy, inf) uint8 arrays to write into (reusing result buffers spares a
        large batch the page faults of freshly allocated memory)."""
        B = FIELD_BYTES[curve]
        k = _u8(k, (-1, B))
        xy = _u8(xy, (-1, 2 * B))
        n = k.shape[0]
        assert xy.shape[0] == n
        if out is None:
            out, inf = np.zeros((n, 2 * B), np.uint8), np.zeros(n, np.uint8)
        else:
            out, inf = out
            assert out.dtype == np.uint8 and out.shape == (n, 2 * B) and out.flags.c_contiguous
            assert inf.dtype == np.uint8 and inf.shape == (n,) and inf.flags.c_contiguous
        self._check(self._lib.ellgpu_mul_var(self._ctx, self._cid(curve), n, k.ctypes.data,
                                             xy.ctypes.data, out.ctypes.data, inf.ctypes.data))
        return out, inf

    def mul_add2(self, curve, k1, p1, k2, p2):
        """k1*P1 + k2*P2; p1=None means P1 = G."""
        B = FIELD_BYTES[curve]
        k1 = _u8(k1, (-1, B))
        n = k1.shape[0]
        k2 = _u8(k2, (n, B))
        p2 = _u8(p2, (n, 2 * B))
        p1p = None
        if p1 is not None:
            p1 = _u8(p1, (n, 2 * B))
            p1p = p1.ctypes.data
        out = np.zeros((n, 2 * B), np.uint8)
        inf = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_mul_add2(self._ctx, self._cid(curve), n, k1.ctypes.data, p1p,
                                              k2.ctypes.data, p2.ctypes.data, out.ctypes.data,
                                              inf.ctypes.data))
        return out, inf

    def ecdsa_verify(self, curve, hashes, r, s, pub, msg_bits=0, status=False):
        """EC#verify per item -> ok, a mask of strictly 0 / 1.  status=True -> (ok, st): st[i] = 2
        (ELLGPU_STATUS_OFF_CURVE) where r and s are in range but pub[i] is not on the curve -- ok[i]
        is 0 there, the safe answer; the reference computes with such keys, run it on those items
        if its answer is wanted -- else 0."""
        B, NB = FIELD_BYTES[curve], ORDER_BYTES[curve]
        hashes = _u8(hashes)
        if hashes.ndim != 2:
            raise ValueError("hashes must be (n, hash_len)")
        n, hash_len = hashes.shape
        r = _u8(r, (n, NB))                     # explicit row counts: a short array must not
        s = _u8(s, (n, NB))                     # make the library read past a buffer
        pub = _u8(pub, (n, 2 * B))
        ok = np.zeros(n, np.uint8)
        st = np.zeros(n, np.uint8) if status else None
        self._check(self._lib.ellgpu_ecdsa_verify(self._ctx, self._cid(curve), n, hashes.ctypes.data,
                                                  hash_len, int(msg_bits), r.ctypes.data,
                                                  s.ctypes.data, pub.ctypes.data, ok.ctypes.data,
                                                  st.ctypes.data if status else None))
        return (ok, st) if status else ok

    def decompress(self, curve, v, odd):
        """pointFromX (short curves, v = x) / pointFromY (ed25519, v = y) -> (xy, ok)"""
        B = FIELD_BYTES[curve]
        v = _u8(v, (-1, B))
        n = v.shape[0]
        odd = _u8(odd, (n,))
        out = np.zeros((n, 2 * B), np.uint8)
        ok = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_decompress(self._ctx, self._cid(curve), n, v.ctypes.data,
                                                odd.ctypes.data, out.ctypes.data, ok.ctypes.data))
        return out, ok

    def decompress_dev(self, curve, v, odd, out_xy, out_ok):
        n = v.shape[0]
        self._check(self._lib.ellgpu_decompress_dev(self._ctx, self._cid(curve), n, v.data_ptr(),
                                                    odd.data_ptr(), out_xy.data_ptr(),
                                                    out_ok.data_ptr(), self._stream()))

    def decode_points(self, curve, enc):
        """decodePoint per row of `enc` (n, enc_len) -> (xy, status); status 0 point,
        1 'Unknown point format', 2 'invalid point', 3 'Assertion failed' (hybrid parity)"""
        B = FIELD_BYTES[curve]
        enc = _u8(enc)
        if enc.ndim != 2:
            raise ValueError("enc must be (n, enc_len)")
        n, enc_len = enc.shape
        out = np.zeros((n, 2 * B), np.uint8)
        st = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_decode_points(self._ctx, self._cid(curve), n, enc.ctypes.data,
                                                   enc_len, out.ctypes.data, st.ctypes.data))
        return out, st

    def decode_points_dev(self, curve, enc, out_xy, out_status):
        n, enc_len = enc.shape
        self._check(self._lib.ellgpu_decode_points_dev(self._ctx, self._cid(curve), n, enc.data_ptr(),
                                                       enc_len, out_xy.data_ptr(),
                                                       out_status.data_ptr(), self._stream()))

    def encode_points(self, curve, xy, compact=False):
        """BasePoint#encode / EDDSA#encodePoint per affine point -> (n, enc_len) bytes"""
        B = FIELD_BYTES[curve]
        xy = _u8(xy, (-1, 2 * B))
        n = xy.shape[0]
        enc_len = 32 if curve == "ed25519" else (1 + B if compact else 1 + 2 * B)
        out = np.zeros((n, enc_len), np.uint8)
        self._check(self._lib.ellgpu_encode_points(self._ctx, self._cid(curve), n, xy.ctypes.data,
                                                   1 if compact else 0, out.ctypes.data))
        return out

    def encode_points_dev(self, curve, xy, compact, out_enc):
        n = xy.shape[0]
        self._check(self._lib.ellgpu_encode_points_dev(self._ctx, self._cid(curve), n, xy.data_ptr(),
                                                       1 if compact else 0, out_enc.data_ptr(),
                                                       self._stream()))

    def validate(self, curve, xy, inf=None, check_order=True):
        """KeyPair#validate per item -> status (0 ok, 1 'Invalid public key', 2 'Public key is
        not a point', 3 'Public key * N != O')"""
        B = FIELD_BYTES[curve]
        xy = _u8(xy, (-1, 2 * B))
        n = xy.shape[0]
        if inf is not None:
            inf = _u8(inf, (n,))
        st = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_validate(self._ctx, self._cid(curve), n, xy.ctypes.data,
                                              inf.ctypes.data if inf is not None else None,
                                              1 if check_order else 0, st.ctypes.data))
        return st

    def validate_dev(self, curve, xy, inf, check_order, out_status):
        n = xy.shape[0]
        self._check(self._lib.ellgpu_validate_dev(self._ctx, self._cid(curve), n, xy.data_ptr(),
                                                  inf.data_ptr() if inf is not None else None,
                                                  1 if check_order else 0, out_status.data_ptr(),
                                                  self._stream()))

    @staticmethod
    def _pack_records(items, stride=None):
        """list of byte strings -> ((n, stride) uint8 array, (n,) uint32 lengths)"""
        n = len(items)
        stride = max([stride or 1] + [len(x) for x in items])
        buf = np.zeros((n, stride), np.uint8)
        lens = np.zeros(n, np.uint32)
        for i, x in enumerate(items):
            buf[i, :len(x)] = np.frombuffer(bytes(x), np.uint8)
            lens[i] = len(x)
        return buf, lens

    def sig_from_der(self, curve, sigs):
        """Signature#_importDER per item (sigs: list of byte strings) -> (r, s, status);
        status 0 parsed, 1 malformed ('Signature without r or s'), 2 wider than n"""
        NB = ORDER_BYTES[curve]
        der, lens = self._pack_records(sigs)
        n = len(sigs)
        r = np.zeros((n, NB), np.uint8)
        s = np.zeros((n, NB), np.uint8)
        st = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_sig_from_der(self._ctx, self._cid(curve), n, der.ctypes.data,
                                                  der.shape[1], lens.ctypes.data, r.ctypes.data,
                                                  s.ctypes.data, st.ctypes.data))
        return r, s, st

    def sig_to_der(self, curve, r, s):
        """Signature#toDER per (r, s) -> list of byte strings"""
        NB = ORDER_BYTES[curve]
        r = _u8(r, (-1, NB))
        n = r.shape[0]
        s = _u8(s, (n, NB))
        stride = 2 * NB + 9
        out = np.zeros((n, stride), np.uint8)
        lens = np.zeros(n, np.uint32)
        self._check(self._lib.ellgpu_sig_to_der(self._ctx, self._cid(curve), n, r.ctypes.data, s.ctypes.data,
                                                out.ctypes.data, stride, lens.ctypes.data))
        return [out[i, :lens[i]].tobytes() for i in range(n)]

    def ecdsa_verify_wire(self, curve, hashes, sigs, pubs, msg_bits=0):
        """EC#verify(msg, DER signature, encoded key) per item.  sigs: list of byte strings (or
        the packed pair _pack_records returns);
        pubs: (n, pub_len) SEC1 encodings -> (ok, err); err 1..3 = decodePoint's status for the
        key, 4 = 'Signature without r or s'"""
        hashes = _u8(hashes)
        if hashes.ndim != 2:
            raise ValueError("hashes must be (n, hash_len)")
        n, hash_len = hashes.shape
        pubs = _u8(pubs)
        if isinstance(sigs, tuple):                   # already packed: ((n, stride) uint8, (n,) uint32)
            der, lens = _u8(sigs[0]), np.ascontiguousarray(sigs[1], np.uint32)
        else:
            der, lens = self._pack_records(sigs)
        if pubs.ndim != 2 or pubs.shape[0] != n or der.shape[0] != n or lens.shape[0] != n:
            raise ValueError("pubs must be (n, pub_len), sigs n byte strings (or packed records + lengths)")
        ok = np.zeros(n, np.uint8)
        err = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_ecdsa_verify_wire(self._ctx, self._cid(curve), n, hashes.ctypes.data,
                                                       hash_len, int(msg_bits), der.ctypes.data, der.shape[1],
                                                       lens.ctypes.data, pubs.ctypes.data, pubs.shape[1],
                                                       ok.ctypes.data, err.ctypes.data))
        return ok, err

    def ecdsa_verify_wire_dev(self, curve, hashes, der, der_len, pubs, out_ok, out_err=None, msg_bits=0):
        n, hash_len = hashes.shape
        self._check(self._lib.ellgpu_ecdsa_verify_wire_dev(
            self._ctx, self._cid(curve), n, hashes.data_ptr(), hash_len, int(msg_bits), der.data_ptr(),
            der.shape[1], der_len.data_ptr(), pubs.data_ptr(), pubs.shape[1], out_ok.data_ptr(),
            out_err.data_ptr() if out_err is not None else None, self._stream()))

    def ecdsa_sign(self, curve, hashes, priv, nonces, canonical=False, msg_bits=0):
        """one pass of EC#sign per item for supplied nonces -> (r, s, recid, ok)"""
        NB = ORDER_BYTES[curve]
        hashes = _u8(hashes)
        n, hash_len = hashes.shape
        priv = _u8(priv, (n, NB))
        nonces = _u8(nonces, (n, NB))
        r = np.zeros((n, NB), np.uint8)
        s = np.zeros((n, NB), np.uint8)
        rec = np.zeros(n, np.uint8)
        ok = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_ecdsa_sign(self._ctx, self._cid(curve), n, hashes.ctypes.data, hash_len,
                                                int(msg_bits), priv.ctypes.data, nonces.ctypes.data,
                                                1 if canonical else 0, r.ctypes.data, s.ctypes.data,
                                                rec.ctypes.data, ok.ctypes.data))
        return r, s, rec, ok

    def ecdsa_sign_dev(self, curve, hashes, priv, nonces, out_r, out_s, out_recid, out_ok, canonical=False,
                       msg_bits=0):
        n, hash_len = hashes.shape
        self._check(self._lib.ellgpu_ecdsa_sign_dev(self._ctx, self._cid(curve), n, hashes.data_ptr(), hash_len,
                                                    int(msg_bits), priv.data_ptr(), nonces.data_ptr(),
                                                    1 if canonical else 0, out_r.data_ptr(), out_s.data_ptr(),
                                                    out_recid.data_ptr(), out_ok.data_ptr(), self._stream()))

    def eddsa_verify(self, msgs, sigs, pubs):
        """ed25519 EdDSA verify.  msgs: list of bytes objects (any lengths) or an (n, len)
        uint8 array; sigs (n, 64), pubs (n, 32) in wire encoding.  -> (ok, err) uint8 arrays"""
        sigs = _u8(sigs, (-1, 64))
        n = sigs.shape[0]
        pubs = _u8(pubs, (n, 32))
        ok = np.zeros(n, np.uint8)
        err = np.zeros(n, np.uint8)
        if isinstance(msgs, np.ndarray) and msgs.ndim == 2:
            m = np.ascontiguousarray(msgs, np.uint8)
            if m.shape[0] != n:
                raise ValueError("msgs has %d rows for %d signatures" % (m.shape[0], n))
            off_p, mlen = None, m.shape[1]
        else:
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum([len(x) for x in msgs])
            m = np.frombuffer(b"".join(bytes(x) for x in msgs) or b"\0", dtype=np.uint8)
            off_p, mlen = off.ctypes.data, 0
            self._keep = off
        self._check(self._lib.ellgpu_eddsa_verify(self._ctx, n, m.ctypes.data, off_p, mlen,
                                                  sigs.ctypes.data, pubs.ctypes.data, ok.ctypes.data,
                                                  err.ctypes.data))
        return ok, err

    def ecdsa_sign_det(self, curve, hashes, priv, canonical=False, msg_bits=0):
        """EC#sign with the reference's own (HmacDRBG, deterministic) nonces -> (r, s, recid, ok)"""
        NB = ORDER_BYTES[curve]
        hashes = _u8(hashes)
        n, hash_len = hashes.shape
        priv = _u8(priv, (n, NB))
        r = np.zeros((n, NB), np.uint8)
        s = np.zeros((n, NB), np.uint8)
        rec = np.zeros(n, np.uint8)
        ok = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_ecdsa_sign_det(self._ctx, self._cid(curve), n, hashes.ctypes.data, hash_len,
                                                    int(msg_bits), priv.ctypes.data, 1 if canonical else 0,
                                                    r.ctypes.data, s.ctypes.data, rec.ctypes.data, ok.ctypes.data))
        return r, s, rec, ok

    def ecdsa_sign_det_dev(self, curve, hashes, priv, out_r, out_s, out_recid, out_ok, canonical=False,
                           msg_bits=0):
        n, hash_len = hashes.shape
        self._check(self._lib.ellgpu_ecdsa_sign_det_dev(self._ctx, self._cid(curve), n, hashes.data_ptr(), hash_len,
                                                        int(msg_bits), priv.data_ptr(), 1 if canonical else 0,
                                                        out_r.data_ptr(), out_s.data_ptr(), out_recid.data_ptr(),
                                                        out_ok.data_ptr(), self._stream()))

    def ecdsa_recover(self, curve, hashes, r, s, recid):
        """EC#recoverPubKey per item -> (xy (n, 2B), status (n,)): 0 point, 1 infinity,
        2 the reference throws, 3 outside the engine's domain (r = 0 or r >= n)"""
        B, NB = FIELD_BYTES[curve], ORDER_BYTES[curve]
        hashes = _u8(hashes)
        n, hash_len = hashes.shape
        r = _u8(r, (n, NB))
        s = _u8(s, (n, NB))
        recid = _u8(recid, (n,))
        xy = np.zeros((n, 2 * B), np.uint8)
        st = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_ecdsa_recover(self._ctx, self._cid(curve), n, hashes.ctypes.data, hash_len,
                                                   r.ctypes.data, s.ctypes.data, recid.ctypes.data,
                                                   xy.ctypes.data, st.ctypes.data))
        return xy, st

    def ecdsa_recover_dev(self, curve, hashes, r, s, recid, out_xy, out_status):
        n, hash_len = hashes.shape
        self._check(self._lib.ellgpu_ecdsa_recover_dev(self._ctx, self._cid(curve), n, hashes.data_ptr(), hash_len,
                                                       r.data_ptr(), s.data_ptr(), recid.data_ptr(),
                                                       out_xy.data_ptr(), out_status.data_ptr(), self._stream()))

    def eddsa_sign(self, msgs, secrets):
        """ed25519 EdDSA sign from 32-byte secrets (EDDSA#sign with keyFromSecret).  msgs as for
        eddsa_verify.  -> (sig (n, 64), pub (n, 32)) uint8 arrays"""
        secrets = _u8(secrets, (-1, 32))
        n = secrets.shape[0]
        sig = np.zeros((n, 64), np.uint8)
        pub = np.zeros((n, 32), np.uint8)
        if isinstance(msgs, np.ndarray) and msgs.ndim == 2:
            m = np.ascontiguousarray(msgs, np.uint8)
            if m.shape[0] != n:
                raise ValueError("msgs has %d rows for %d signatures" % (m.shape[0], n))
            off_p, mlen = None, m.shape[1]
            if m.size == 0:
                m = np.zeros(1, np.uint8)
        else:
            off = np.zeros(n + 1, np.uint64)
            off[1:] = np.cumsum([len(x) for x in msgs])
            m = np.frombuffer(b"".join(bytes(x) for x in msgs) or b"\0", dtype=np.uint8)
            off_p, mlen = off.ctypes.data, 0
        self._check(self._lib.ellgpu_eddsa_sign(self._ctx, n, secrets.ctypes.data, m.ctypes.data, off_p, mlen,
                                                sig.ctypes.data, pub.ctypes.data))
        return sig, pub

    def eddsa_sign_dev(self, secrets, msgs, msg_len, out_sig, out_pub=None, msg_off=None):
        n = secrets.shape[0]
        self._check(self._lib.ellgpu_eddsa_sign_dev(
            self._ctx, n, secrets.data_ptr(), msgs.data_ptr(), None if msg_off is None else msg_off.data_ptr(),
            int(msg_len), out_sig.data_ptr(), None if out_pub is None else out_pub.data_ptr(), self._stream()))

    def eddsa_verify_dev(self, msgs, msg_len, sigs, pubs, out_ok, out_err=None, msg_off=None):
        n = sigs.shape[0]
        self._check(self._lib.ellgpu_eddsa_verify_dev(
            self._ctx, n, msgs.data_ptr(), None if msg_off is None else msg_off.data_ptr(), int(msg_len),
            sigs.data_ptr(), pubs.data_ptr(), out_ok.data_ptr(),
            None if out_err is None else out_err.data_ptr(), self._stream()))

    def point_add(self, curve, xy1, xy2, inf1=None, inf2=None):
        """Point#add per pair of affine points (infinity through the optional flags) -> (xy, inf)"""
        B = FIELD_BYTES[curve]
        xy1 = _u8(xy1, (-1, 2 * B))
        n = xy1.shape[0]
        xy2 = _u8(xy2, (n, 2 * B))
        inf1 = None if inf1 is None else _u8(inf1, (n,))
        inf2 = None if inf2 is None else _u8(inf2, (n,))
        out = np.zeros((n, 2 * B), np.uint8)
        inf = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_point_add(self._ctx, self._cid(curve), n, xy1.ctypes.data,
                                               None if inf1 is None else inf1.ctypes.data, xy2.ctypes.data,
                                               None if inf2 is None else inf2.ctypes.data,
                                               out.ctypes.data, inf.ctypes.data))
        return out, inf

    def point_add_dev(self, curve, xy1, xy2, out_xy, out_inf, inf1=None, inf2=None):
        n = xy1.shape[0]
        self._check(self._lib.ellgpu_point_add_dev(self._ctx, self._cid(curve), n, xy1.data_ptr(),
                                                   None if inf1 is None else inf1.data_ptr(), xy2.data_ptr(),
                                                   None if inf2 is None else inf2.data_ptr(),
                                                   out_xy.data_ptr(), out_inf.data_ptr(), self._stream()))

    def ecdh_derive(self, curve, priv, pub_xy):
        """KeyPair#derive per item (ec/key.js:102-107): pub.validate() then pub.mul(priv).getX()
        -> (x, status); status 0 shared secret, 1 'public point not validated', 2 the product is
        the point at infinity (getX throws in the reference).  Two launches: the curve-equation
        kernel and the variable-base ladder."""
        B = FIELD_BYTES[curve]
        pub_xy = _u8(pub_xy, (-1, 2 * B))
        n = pub_xy.shape[0]
        priv = _u8(priv, (n, B))
        st = self.validate(curve, pub_xy, check_order=False)
        xy, inf = self.mul_var(curve, priv, pub_xy)
        status = np.where(st != 0, 1, np.where(inf != 0, 2, 0)).astype(np.uint8)
        x = np.ascontiguousarray(xy[:, :B])
        x[status != 0] = 0
        return x, status

    def x25519(self, k, x):
        k = _u8(k, (-1, 32))
        x = _u8(x, (-1, 32))
        n = k.shape[0]
        out = np.zeros((n, 32), np.uint8)
        inf = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_x25519_ladder(self._ctx, n, k.ctypes.data, x.ctypes.data,
                                                   out.ctypes.data, inf.ctypes.data))
        return out, inf

    # ---- device buffers (torch CUDA uint8 tensors) ------------------------------
    def _stream(self):
        """the current torch stream OF THIS CONTEXT'S DEVICE (not of torch's current device)"""
        import torch
        cur = torch.cuda.current_stream(self.device)
        if cur.cuda_stream:
            return ctypes.c_void_p(cur.cuda_stream)
        # torch's default stream is the NULL handle, which the C ABI reads as "the context's own
        # (non-blocking) stream": run there, ordered after the default stream's work so far, and
        # let the default stream wait for the call's work afterwards (_check)
        if self._own is None:
            self._own = torch.cuda.ExternalStream(self._lib.ellgpu_ctx_stream(self._ctx), device=self.device)
        self._own.wait_stream(cur)
        self._pending_default = cur
        return ctypes.c_void_p(self._own.cuda_stream)

    def mul_fixed_dev(self, curve, k, out_xy, out_inf):
        n = k.shape[0]
        self._check(self._lib.ellgpu_mul_fixed_dev(self._ctx, self._cid(curve), n, k.data_ptr(),
                                                   out_xy.data_ptr(), out_inf.data_ptr(),
                                                   self._stream()))

    def mul_var_dev(self, curve, k, xy, out_xy, out_inf):
        n = k.shape[0]
        self._check(self._lib.ellgpu_mul_var_dev(self._ctx, self._cid(curve), n, k.data_ptr(),
                                                 xy.data_ptr(), out_xy.data_ptr(),
                                                 out_inf.data_ptr(), self._stream()))

    def mul_add2_dev(self, curve, k1, p1, k2, p2, out_xy, out_inf):
        n = k1.shape[0]
        self._check(self._lib.ellgpu_mul_add2_dev(self._ctx, self._cid(curve), n, k1.data_ptr(),
                                                  None if p1 is None else p1.data_ptr(),
                                                  k2.data_ptr(), p2.data_ptr(), out_xy.data_ptr(),
                                                  out_inf.data_ptr(), self._stream()))

    def ecdsa_verify_dev(self, curve, hashes, r, s, pub, out_ok, msg_bits=0, out_status=None):
        """out_ok: uint8 CUDA tensor (n), strictly 0 / 1; out_status (optional, same shape): 2 where
        the key is not on the curve (out_ok 0 there), else 0 -- see ecdsa_verify"""
        n, hash_len = hashes.shape
        self._check(self._lib.ellgpu_ecdsa_verify_dev(self._ctx, self._cid(curve), n,
                                                      hashes.data_ptr(), hash_len, int(msg_bits),
                                                      r.data_ptr(), s.data_ptr(), pub.data_ptr(),
                                                      out_ok.data_ptr(),
                                                      None if out_status is None else out_status.data_ptr(),
                                                      self._stream()))

    def x25519_derive(self, k, x):
        """KeyPair#derive on curve25519 per item: validate (is x an abscissa of the curve?) and the
        ladder in one call -> (out_x (n, 32), status (n,)): 0 shared secret, 1 x has no point (the
        reference throws 'Assertion failed' out of its square root), 2 the product is infinity"""
        k = _u8(k, (-1, 32))
        n = k.shape[0]
        x = _u8(x, (n, 32))
        out = np.zeros((n, 32), np.uint8)
        st = np.zeros(n, np.uint8)
        self._check(self._lib.ellgpu_x25519_derive(self._ctx, n, k.ctypes.data, x.ctypes.data, out.ctypes.data,
                                                   st.ctypes.data))
        return out, st

    def x25519_dev(self, k, x, out_x, out_inf):
        n = k.shape[0]
        self._check(self._lib.ellgpu_x25519_ladder_dev(self._ctx, n, k.data_ptr(), x.data_ptr(),
                                                       out_x.data_ptr(), out_inf.data_ptr(),
                                                       self._stream()))

    def set_timing(self, on=True):
        self._check(self._lib.ellgpu_ctx_set_timing(self._ctx, 1 if on else 0))

    def get_timing(self):
        """-> {kernel name: (launches, total_ms)} since set_timing(True)"""
        buf = ctypes.create_string_buffer(8192)
        rc = self._lib.ellgpu_ctx_get_timing(self._ctx, buf, 8192)
        if rc < 0:
            self._check(rc)
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def probe_valu(self, kind, blocks, iters):
        ms = ctypes.c_double()
        ops = ctypes.c_double()
        self._check(self._lib.ellgpu_probe_valu(self._ctx, kind, blocks, iters, ctypes.byref(ms),
                                                ctypes.byref(ops)))
        return ms.value, ops.value
