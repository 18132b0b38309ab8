/*
 * ellgpu.h -- C ABI of libellgpu.so, the MI355X (gfx950) batched elliptic-curve
 * scalar-multiplication engine that stands behind indutny/elliptic's
 * Point.mul / mulAdd / jmulAdd / ec.verify hot path.
 *
 * The reference (pure JavaScript, /root/reference) has NO FFI or plugin
 * interface; its only seam is a set of synchronous, one-item-per-call
 * prototype methods.  Each entry point below is the batched form of one of
 * them -- the function an N-API addon binds so that a patched
 * `elliptic` routes that method here:
 *
 *   ellgpu_mul_fixed     <- Point#mul on a precomputed G  -> BaseCurve#_fixedNafMul
 *                           lib/elliptic/curve/short.js:422-427, base.js:52-84
 *                           (edwards.js:362-365 for ed25519)
 *   ellgpu_mul_var       <- Point#mul, variable base      -> _endoWnafMulAdd / _wnafMul
 *                           short.js:428-431, 218-249; base.js:86-126; edwards.js:366
 *   ellgpu_mul_add2      <- Point#mulAdd / jmulAdd        -> _wnafMulAdd
 *                           short.js:434-450; base.js:128-253; edwards.js:369-375
 *   ellgpu_ecdsa_verify  <- EC#verify after key/signature decoding
 *                           lib/elliptic/ec/index.js:188-229 (_truncateToN :81-108,
 *                           JPoint#eqXToP short.js:908-925)
 *   ellgpu_x25519_ladder <- mont Point#mul + getX         lib/elliptic/curve/mont.js:130-178
 *
 * Conventions
 *   - Plain pointers and sizes only.  All integers (scalars, coordinates, r, s)
 *     are fixed-width BIG-ENDIAN byte strings, item-major, exactly what
 *     BN#toArray('be', len) / SEC1 04||x||y give (base.js:298-306): width =
 *     ellgpu_curve_field_bytes() for scalars and coordinates,
 *     ellgpu_curve_order_bytes() for r and s.  Scalars are NOT reduced mod n by
 *     the callee, matching the reference (any value of that width is accepted).
 *   - Points are affine x||y.  A result at infinity is reported through
 *     out_inf[i] = 1 with its x||y zeroed (the reference returns
 *     curve.point(null, null), short.js:253-256; for ed25519 the identity (0,1)
 *     is an ordinary point and out_inf mirrors Point#isInfinity()).
 *   - DOMAIN: input points must lie ON the curve, and the callee CHECKS it, per item.
 *     The reference does not validate on this path (ec/index.js:192 keyFromPublic,
 *     ec/key.js:27-35, curve/short.js:422-432, curve/edwards.js:362-367): it runs its
 *     formulas on any (x, y), and off the curve their outcome depends on the exact order of
 *     its operations (wNAF / JSF digits, its GLV split, the window of G's shipped table) --
 *     there is no group law to agree on.  The engine's ladders are different by design, so
 *     an item with an operand that is not on the curve is OUTSIDE THE ENGINE'S DOMAIN and is
 *     reported as such, never answered with a guess:
 *         ELLGPU_STATUS_OFF_CURVE (= 2)  in out_inf[i] of ellgpu_mul_var / _mul_add2 (x||y
 *                                        zeroed), in out_status[i] of ellgpu_ecdsa_verify
 *                                        (out_err[i] = 5 for ellgpu_ecdsa_verify_wire)
 *     A verdict array (out_ok) is a MASK, strictly 0 / 1, and it is 0 for such an item: a caller
 *     that reads it as a boolean never accepts a signature over a key that is no curve point.
 *     Callers that want the reference's answer for such an item run the reference on it
 *     (elliptic_amd/js/index.js install() does exactly that; tests/golden/offcurve_*.json
 *     pins what the reference answers).  ellgpu_point_add is the one exception: a single
 *     chord / tangent formula, the reference's own, defined and equal off the curve too.
 *     x-only ellgpu_x25519_ladder: every x is on the curve or its twist and the ladder is the
 *     same function of (k, x) on both -- no such status.  Coordinates >= p are reduced mod p
 *     first, as curve.point() does.  Infinity as an INPUT is not representable; the JS layer
 *     short-circuits it exactly as short.js:424-425 does.
 *   - Per-item failures (bad r/s range, result at infinity) are per-item
 *     status bytes, never errors: EC#verify returns false, it does not throw
 *     (ec/index.js:199-202,222-223).
 *   - Every function returns 0 on success or a negative ELLGPU_E_* code;
 *     ellgpu_last_error() gives a thread-local message.  There is NO CPU
 *     fallback: without a usable gfx950 device ellgpu_ctx_create fails.
 *   - A context is bound to one device and owns its HIP streams, staging buffers and scratch
 *     arenas.  Entry points on ONE context take turns: every call holds the context's mutex --
 *     host-buffer calls until their results are back, *_dev calls while they enqueue -- so a
 *     second host thread (the N-API addon's libuv worker beside the JS thread) waits instead
 *     of overwriting the first one's staged inputs.  Throughput comes from one context per
 *     GPU (one process / thread per GPU, see DESIGN.md multi-GPU), not from threads inside a
 *     context.  Host-buffer entry points are synchronous.
 *     The *_dev entry points take DEVICE pointers, enqueue on the given
 *     hipStream_t (passed as void*, NULL = the context's own non-blocking stream, which
 *     ellgpu_ctx_stream returns so that a caller can order other streams against it) and return
 *     without synchronising; outputs are valid once that stream is.  The scratch belongs to the
 *     context, which keeps TWO arenas and gives a *_dev call the arena of ITS stream: calls issued
 *     alternately on two streams run side by side (two passes in flight -- the latency-bound front
 *     of one beside the issue-bound ladder of the other: 131 072 verifies 1.27 -> 1.08 ms per pass,
 *     2^20 8.08 -> 7.75 ms), without host synchronisation in between.  A call on a third stream
 *     takes the arena used longest ago and is ordered, on the DEVICE, behind the event its previous
 *     user recorded -- more streams serialise pairwise, they never race.  Host-buffer calls use
 *     both arenas and wait for whatever the *_dev calls left in flight.  (ELLGPU_DEV_LANES=1,
 *     read when the context is created: one arena for all *_dev calls, i.e. they serialise.)
 *     EXCEPTION: on a user-defined curve (ellgpu_curve_define_short / _edwards) a *_dev call is
 *     SYNCHRONOUS: the curve's parameter block lives in the device's constant memory, one block
 *     per device, so such a call takes a per-device lock, uploads the block, and waits for its
 *     own device work before the lock is released -- it blocks the host until its stream has
 *     drained, and serialises against other contexts' user-defined-curve calls on that device.
 */
#ifndef ELLGPU_H
#define ELLGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever a prototype below changes (0x000200: ellgpu_ecdsa_verify and
 * ellgpu_ecdsa_verify_dev gained out_status and their out_ok became a strict 0 / 1 mask;
 * ellgpu_ctx_defer / ellgpu_ctx_collect).  Both loaders in this repo -- elliptic_amd/_lib.py and
 * elliptic_amd/js/ellgpu_napi.c -- refuse a library whose ellgpu_version() differs from the value
 * they were written against: a stale binding would pass garbage for the added arguments. */
#define ELLGPU_VERSION 0x000200

/* curve ids (names are the reference's preset names, lib/elliptic/curves.js) */
#define ELLGPU_SECP256K1 0
#define ELLGPU_P192 1
#define ELLGPU_P224 2
#define ELLGPU_P256 3
#define ELLGPU_P384 4
#define ELLGPU_P521 5
#define ELLGPU_ED25519 6
#define ELLGPU_CURVE25519 7

#define ELLGPU_OK 0
#define ELLGPU_E_NODEVICE (-1)  /* no gfx950 device / HIP runtime failure at init */
#define ELLGPU_E_ARG (-2)       /* bad argument (unknown curve, NULL pointer, unsupported width) */
#define ELLGPU_E_HIP (-3)       /* HIP runtime error during the call */
#define ELLGPU_E_NOMEM (-4)     /* device allocation failed */
#define ELLGPU_E_UNSUPPORTED (-5) /* operation not defined for this curve (e.g. mulAdd on curve25519, mont.js:155) */

/* per-item status in out_inf / out_status: an operand is not on the curve (see DOMAIN above) */
#define ELLGPU_STATUS_OFF_CURVE 2

typedef struct ellgpu_ctx ellgpu_ctx;

int ellgpu_version(void);
/* 16 hex digits: sha256 over the (repo-relative name, bytes) of every source file this library
 * was compiled from + the compiler flags (elliptic_amd/build.py::source_digest, passed to the
 * compiler as -DELLGPU_SOURCE_DIGEST).  Provenance of measurements and of the smoke gate is read
 * from the loaded binary through this call, never from a sidecar file.  No reference
 * counterpart (package.json "version" is the nearest thing). */
const char* ellgpu_source_digest(void);
const char* ellgpu_last_error(void);

int ellgpu_curve_id(const char* name);       /* "secp256k1", "p192", ... ; -1 if unknown */
int ellgpu_curve_field_bytes(int curve);     /* 32, 24, 28, 32, 48, 66, 32, 32 */
int ellgpu_curve_order_bytes(int curve);     /* byte length of n */

int ellgpu_device_count(void);               /* number of visible HIP devices, <0 on error */
int ellgpu_ctx_create(int device, ellgpu_ctx** out);
/* Multi-GPU (the reference has no counterpart -- it is one JS thread):
 * a GROUP is a context over `ndev` devices (devices[i] = HIP ordinal; an ordinal may repeat).
 * ellgpu_mul_fixed / _mul_var / _mul_add2 / _ecdsa_verify called on it cut the batch into ndev
 * contiguous shards [n*i/ndev, n*(i+1)/ndev), drive every device from its own host thread
 * (own streams, scratch and replicated tables) and let each device copy its results straight
 * into the caller's buffers: items are independent, so there is no device-to-device traffic
 * and no collective.  Every other entry point called on a group runs on its first device.
 * Destroy with ellgpu_ctx_destroy. */
int ellgpu_group_create(const int* devices, int ndev, ellgpu_ctx** out);
int ellgpu_group_size(const ellgpu_ctx* ctx);          /* ndev of a group, 1 for a plain context */
void ellgpu_ctx_destroy(ellgpu_ctx* ctx);
int ellgpu_ctx_synchronize(ellgpu_ctx* ctx);
/* the context's own stream (a hipStream_t; a group: its first member's): what a NULL `stream`
 * argument of the *_dev entry points means */
void* ellgpu_ctx_stream(ellgpu_ctx* ctx);
/* The SPLIT form of a small host-buffer call.  The reference's API is one item per synchronous
 * call (lib/elliptic/ec/index.js:188-229, curve/short.js:422-432) and takes a point's precomputed
 * tables on trust (curve/base.js:52-126); a patched call must find them unchanged on EVERY call
 * (elliptic_amd/js/index.js, INTEGRATION.md "what install() treats as immutable") -- a walk over a
 * few hundred table entries on the host.  ellgpu_ctx_defer(ctx) arms the context: its NEXT
 * host-buffer call, if it is one of the few-item calls that travel through the context's pinned
 * buffer (<= 256 KB of operands and results; not a group, not a user-defined curve), returns as
 * soon as its copies and kernels are enqueued, and the caller's result buffers are written by
 * ellgpu_ctx_collect(ctx), which waits for the device and returns the call's status -- the host
 * validates while the device computes.  The result buffers must stay valid until then.  A call
 * that cannot be deferred runs to completion as usual and ellgpu_ctx_collect has nothing to do;
 * any other entry point on the context completes a pending call first.  collect also disarms.
 * Only the ARMING THREAD's next call is deferred: a call another host thread makes in between (the
 * N-API worker running a Promise-form batch beside the JS thread) runs to completion as usual. */
int ellgpu_ctx_defer(ellgpu_ctx* ctx);
int ellgpu_ctx_collect(ellgpu_ctx* ctx);

/* User-defined short Weierstrass curve y^2 = x^3 + a x + b over an odd prime p < 2^256 -- the
 * reference's `new elliptic.curve.short({p, a, b, ...})` (lib/elliptic/curve/short.js:11-24) with
 * parameters that are none of the presets; p, a, b are 32-byte big-endian (a, b reduced mod p).
 * Registers the curve with the context (a group: with every member) and returns its id
 * (>= ELLGPU_CURVE_CUSTOM0, at most 16 per context; defining the same parameters twice returns the
 * same id).  The id is valid for ellgpu_mul_var / _mul_add2 (both points given) / _point_add and
 * their _dev forms, with 32-byte scalars and coordinates whatever p's size: Point#mul, mulAdd /
 * jmulAdd and Point#add with the generic-a doubling of JPoint#_dbl / dblp (short.js:802-830,
 * 605-654) on the device.  Every other entry point answers ELLGPU_E_UNSUPPORTED for it (the
 * reference's own JavaScript keeps serving those).  The primality of p is not checked, as the
 * reference does not check it either -- but field inversion here is Fermat's a^(p-2), so for a
 * composite p results differ from the reference's (its BN#invm is an extended Euclid); callers
 * that cannot vouch for p keep such curves on the reference (the JS layer runs a Miller-Rabin
 * test before it registers a curve).  On a group, either every member registers the curve or
 * none does. */
#define ELLGPU_CURVE_CUSTOM0 16
int ellgpu_curve_define_short(ellgpu_ctx* ctx, const uint8_t* p, const uint8_t* a, const uint8_t* b,
                              int* out_curve);
/* User-defined (twisted) Edwards curve a x^2 + y^2 = 1 + d x^2 y^2 (c = 1) over an odd prime
 * p < 2^256 -- `new elliptic.curve.edwards({p, a, c: 1, d, ...})` (lib/elliptic/curve/edwards.js:
 * 11-31) with parameters that are not ed25519's.  Same id space, widths (32 bytes) and entry
 * points as ellgpu_curve_define_short; points are affine x || y, the identity is the ordinary
 * point (0, 1) (out_inf stays 0; a set inf flag on input of ellgpu_point_add means (0, 1)).
 * Point#mul / mulAdd / Point#add with the projective formulas the reference uses for a != -1
 * (_projDbl / _projAdd, edwards.js:207-266, 311-348) and, for a = -1, the same group law its
 * extended formulas compute.  The addition law is complete when a is a square and d is not;
 * on other curves exceptional inputs give what the formulas give (a zero denominator comes out
 * as (0, 0)). */
int ellgpu_curve_define_edwards(ellgpu_ctx* ctx, const uint8_t* p, const uint8_t* a, const uint8_t* d,
                                int* out_curve);

/* ---- host-buffer entry points (what the N-API addon binds) -------------- */

/* out[i] = k[i] * G */
int ellgpu_mul_fixed(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k,
                     uint8_t* out_xy, uint8_t* out_inf);
/* out[i] = k[i] * P[i];  out_inf[i] = 0 finite, 1 infinity, 2 P[i] is not on the curve */
int ellgpu_mul_var(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k,
                   const uint8_t* in_xy, uint8_t* out_xy, uint8_t* out_inf);
/* out[i] = k1[i] * P1[i] + k2[i] * P2[i];  p1_xy == NULL means P1 = G;
 * out_inf[i] = 2 when P1[i] or P2[i] is not on the curve */
int ellgpu_mul_add2(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k1,
                    const uint8_t* p1_xy, const uint8_t* k2, const uint8_t* p2_xy,
                    uint8_t* out_xy, uint8_t* out_inf);
/* out_ok[i] = EC#verify(hash[i], {r[i], s[i]}, pub[i]): strictly 1 / 0.
 * out_status (may be NULL): out_status[i] = 2 (ELLGPU_STATUS_OFF_CURVE) when r and s are in range
 * but pub[i] is not on the curve, else 0.  out_ok[i] is 0 for such an item -- the safe answer;
 * the reference goes on to compute with such a key (and can answer true,
 * tests/golden/offcurve_*.json), the engine does not guess: a caller that wants the reference's
 * own answer runs the reference on the items out_status marks.
 * hash: n x hash_len bytes, the message digest exactly as the caller would
 * pass it to EC#verify as an array (its length, not its value, drives the
 * truncation: ec/index.js:86-96).  msg_bits = 0 means hash_len*8; otherwise it
 * is options.msgBitLength (ec/index.js:97-100).  Requires
 * hash_len*8 - max(0, msg_bits - n.bitLength()) <= 8*order_bytes rounded up to
 * 32 bits. */
int ellgpu_ecdsa_verify(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash,
                        int hash_len, int msg_bits, const uint8_t* r, const uint8_t* s,
                        const uint8_t* pub_xy, uint8_t* out_ok, uint8_t* out_status);
/* x-only Montgomery ladder on curve25519: out_x[i] = x(k[i] * (in_x[i], .));
 * out_inf[i] = 1 when the result is the point at infinity (Z == 0), where the
 * reference's getX() would throw. */
int ellgpu_x25519_ladder(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                         uint8_t* out_x, uint8_t* out_inf);
/* KeyPair#derive on curve25519 (lib/elliptic/ec/key.js:102-107): pub.validate() -- MontCurve#validate,
 * curve/mont.js:23-32: is x^3 + a x^2 + x a square? -- and pub.mul(priv).getX() in ONE call, the
 * validity test (Euler's criterion) on lanes / a wave of its own beside the ladder.  out_status[i]:
 * 0 out_x[i] is the shared secret; 1 in_x[i] is no abscissa of the curve (the reference's validate
 * does not answer false there: bn.js's Tonelli-Shanks square root THROWS 'Assertion failed' on a
 * non-residue, dist/elliptic.js:7242-7302 -- install() runs the reference's own derive on such an
 * item so that the caller sees that exception); 2 the product is the point at infinity. */
int ellgpu_x25519_derive(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                         uint8_t* out_x, uint8_t* out_status);

/* Point decompression.  Short curves: ShortCurve#pointFromX
 * (lib/elliptic/curve/short.js:187-204) -- v[i] is the abscissa, the result has
 * y = sqrt(x^3 + a x + b) with parity odd[i] (SEC1 02/03 prefixes: odd = prefix & 1,
 * base.js:283-289).  ed25519: EdwardsCurve#pointFromY (edwards.js:71-97) -- v[i] is y, the
 * result has the x of parity odd[i] & 1; with bit 1 of odd[i] set (odd[i] = 2 or 3) the item is
 * EdwardsCurve#pointFromX instead (edwards.js:50-69): v[i] is x, the result has the y of parity
 * odd[i] & 1.  out_ok[i] = 0 (x||y zeroed) where the reference throws: 'invalid point', or --
 * over p224 (p = 1 mod 4), whose square root is bn.js's Tonelli-Shanks loop, run here on a
 * fixed schedule -- 'Assertion failed' out of that loop (dist/elliptic.js:7296). */
int ellgpu_decompress(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* v, const uint8_t* odd,
                      uint8_t* out_xy, uint8_t* out_ok);
int ellgpu_decompress_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* v, const uint8_t* odd,
                          uint8_t* out_xy, uint8_t* out_ok, void* stream);

/* Point codecs ("SEC1 & ed25519 codecs").
 * ellgpu_decode_points: n encodings of enc_len bytes each -> affine x||y (big-endian, reduced
 * mod p) + a status byte per item.
 *   short curves: BaseCurve#decodePoint (lib/elliptic/curve/base.js:270-293).  enc_len = 1+2B
 *     with prefix 04 / 06 / 07 (uncompressed, hybrid), or 1+B with prefix 02 / 03 (compressed,
 *     -> pointFromX).  status 0 = point; 1 = 'Unknown point format' (prefix does not fit
 *     enc_len); 2 = 'invalid point' (no y for that x); 3 = 'Assertion failed' (hybrid prefix
 *     contradicts y's last bit, base.js:279-282).  As in the reference an uncompressed point
 *     is NOT checked against the curve equation -- that is ellgpu_validate.  A compressed
 *     p224 encoding without a y has status 3, not 2: the reference's 'Assertion failed' out of
 *     bn.js's Tonelli-Shanks loop (see ellgpu_decompress).
 *   ed25519: EDDSA#decodePoint (lib/elliptic/eddsa/index.js:99-109), enc_len = 32: little-endian
 *     y with x's parity in the top bit.  status 0 / 2.
 * ellgpu_encode_points: affine x||y -> BasePoint#encode (base.js:295-311): compact = 0 gives
 *   04||x||y (1+2B bytes per item), compact != 0 gives 02/03||x (1+B bytes); ed25519:
 *   EDDSA#encodePoint (eddsa/index.js:94-98), 32 bytes per item, `compact` ignored.  The point
 *   at infinity has no encoding (the reference throws); callers filter on their inf flags.
 * ellgpu_validate: KeyPair#validate (lib/elliptic/ec/key.js:41-52) per item.  status 0 =
 *   {result: true}; 1 = 'Invalid public key' (inf[i] != 0; inf may be NULL); 2 = 'Public key is
 *   not a point' (ShortCurve#validate short.js:205-216 / EdwardsCurve#validate
 *   edwards.js:99-112); 3 = 'Public key * N != O' (only evaluated when check_order != 0: one
 *   variable-base n*P per item through the same ladder as ellgpu_mul_var). */
int ellgpu_decode_points(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* enc, size_t enc_len,
                         uint8_t* out_xy, uint8_t* out_status);
int ellgpu_decode_points_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* enc, size_t enc_len,
                             uint8_t* out_xy, uint8_t* out_status, void* stream);
int ellgpu_encode_points(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, int compact,
                         uint8_t* out_enc);
int ellgpu_encode_points_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, int compact,
                             uint8_t* out_enc, void* stream);
int ellgpu_validate(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf,
                    int check_order, uint8_t* out_status);
int ellgpu_validate_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy, const uint8_t* inf,
                        int check_order, uint8_t* out_status, void* stream);

/* Affine point addition: Point#add (lib/elliptic/curve/short.js:365-392, with Point#dbl
 * :394-412 where the operands are equal; lib/elliptic/curve/edwards.js:350-360 -> _extAdd
 * :279-309 for ed25519).  inf1 / inf2 (either may be NULL) flag operands that are the point at
 * infinity (the identity (0, 1) on ed25519); out_inf as for ellgpu_mul_var.  One launch of the
 * chord / tangent formula + the batched normalization, where the reference inverts once per
 * addition.  The formulas are the reference's own, so off-curve operands give the reference's
 * coordinates as well (add_*.json, offcurve_ed25519.json). */
int ellgpu_point_add(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy1, const uint8_t* inf1,
                     const uint8_t* xy2, const uint8_t* inf2, uint8_t* out_xy, uint8_t* out_inf);
int ellgpu_point_add_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* xy1, const uint8_t* inf1,
                         const uint8_t* xy2, const uint8_t* inf2, uint8_t* out_xy, uint8_t* out_inf,
                         void* stream);

/* Signature DER codec (lib/elliptic/ec/signature.js) and EC#verify on wire formats.
 * ellgpu_sig_from_der: Signature#_importDER (signature.js:83-147, with getLength :30-59) of n
 *   records of `stride` bytes, the first der_len[i] of each being the signature.  out_r / out_s
 *   are order_bytes wide, big-endian.  status 0 = parsed; 1 = the reference's `return false`
 *   (its constructor then throws 'Signature without r or s': bad tags, indefinite / overlong /
 *   non-minimal lengths, negative or zero-padded integers, trailing bytes); 2 = well-formed but r
 *   or s is wider than order_bytes (>= n, EC#verify answers false).  r, s are zeroed unless 0.
 * ellgpu_sig_to_der: Signature#toDER (signature.js:149-176) of (r, s); out_len[i] bytes are
 *   written at out_der + i * stride, the rest of the record is zeroed; stride >= 2 *
 *   order_bytes + 9.  s = 0 (where the reference does not terminate) gives out_len 0.
 * ellgpu_ecdsa_verify_wire: EC#verify(msg, der, key) (lib/elliptic/ec/index.js:188-229) with
 *   key = n encodings of pub_len bytes as for ellgpu_decode_points (keyFromPublic ->
 *   decodePoint) and DER signatures as above; point decoding, DER parsing, the double-scalar
 *   multiplication and the x == r test all run on the device.  out_ok = 0/1; out_err (may be
 *   NULL) names the exception the reference throws, in its order: 1..3 = decodePoint's status
 *   for the key, 4 = 'Signature without r or s'; out_ok is 0 wherever out_err is not 0.
 *   out_err 5 is no exception: an uncompressed key that is not on the curve with r, s in range
 *   (ELLGPU_STATUS_OFF_CURVE) -- hand the item to the reference if its answer is wanted. */
int ellgpu_sig_from_der(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* der, size_t stride,
                        const uint32_t* der_len, uint8_t* out_r, uint8_t* out_s, uint8_t* out_status);
int ellgpu_sig_from_der_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* der, size_t stride,
                            const uint32_t* der_len, uint8_t* out_r, uint8_t* out_s, uint8_t* out_status,
                            void* stream);
int ellgpu_sig_to_der(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* r, const uint8_t* s,
                      uint8_t* out_der, size_t stride, uint32_t* out_len);
int ellgpu_sig_to_der_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* r, const uint8_t* s,
                          uint8_t* out_der, size_t stride, uint32_t* out_len, void* stream);
int ellgpu_ecdsa_verify_wire(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                             int msg_bits, const uint8_t* der, size_t der_stride, const uint32_t* der_len,
                             const uint8_t* pub_enc, size_t pub_len, uint8_t* out_ok, uint8_t* out_err);
int ellgpu_ecdsa_verify_wire_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                                 int msg_bits, const uint8_t* der, size_t der_stride,
                                 const uint32_t* der_len, const uint8_t* pub_enc, size_t pub_len,
                                 uint8_t* out_ok, uint8_t* out_err, void* stream);

/* ECDSA sign for caller-supplied nonces: one pass of EC#sign's loop per item
 * (lib/elliptic/ec/index.js:153-185): k = _truncateToN(nonce, true), R = k*G (fixed-base comb),
 * r = R.x mod n, s = k^-1 (z + r d) mod n (k^-1 batched), recoveryParam (:174-175), and with
 * canonical != 0 the low-s form (:178-181).  The nonce itself (HMAC-DRBG / RFC 6979, ec/index.js
 * :143-155) is hashing and stays with the caller.  hash / hash_len / msg_bits as for verify;
 * priv and nonces: n x order_bytes big-endian.  out_ok[i] = 0 (r, s zeroed) where the reference
 * would go on to its next nonce: k <= 1, k >= n-1, k*G at infinity, r = 0 or s = 0. */
int ellgpu_ecdsa_sign(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len, int msg_bits,
                      const uint8_t* priv, const uint8_t* nonces, int canonical, uint8_t* out_r,
                      uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok);
int ellgpu_ecdsa_sign_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                          int msg_bits, const uint8_t* priv, const uint8_t* nonces, int canonical,
                          uint8_t* out_r, uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok,
                          void* stream);

/* EdDSA (ed25519) verify: out_ok[i] = EDDSA#verify(msg_i, sig_i, pub_i),
 * lib/elliptic/eddsa/index.js:52-63 -- S < n, h = SHA-512(R || A || M) mod n (hashInt :65-70),
 * accept iff R + h*A == S*G.  Wire formats as the reference takes them: sig = R || S (64 bytes,
 * little-endian encodings, eddsa/signature.js), pub = A (32 bytes), raw message bytes.
 * Messages are concatenated in `msgs`; msg_off = n+1 byte offsets (message i is
 * msgs[msg_off[i] .. msg_off[i+1])), or NULL for a uniform length msg_len.
 * out_err (may be NULL): 1 where the reference THROWS instead of returning -- R or A does not
 * decode to a curve point (decodePoint -> pointFromY, eddsa/index.js:99-109); out_ok is 0 there. */
int ellgpu_eddsa_verify(ellgpu_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off,
                        size_t msg_len, const uint8_t* sigs, const uint8_t* pubs, uint8_t* out_ok,
                        uint8_t* out_err);
int ellgpu_eddsa_verify_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* msg_off,
                            size_t msg_len, const uint8_t* sigs, const uint8_t* pubs,
                            uint8_t* out_ok, uint8_t* out_err, void* stream);

/* EC#sign with its own nonces -- deterministic signatures, lib/elliptic/ec/index.js:110-186:
 * HmacDRBG (hmac-drbg 1.0.1) over the curve's hash (SHA-256; SHA-384 for p384, SHA-512 for p521;
 * lib/elliptic/curves.js `hash:`), entropy = the private key, nonce = the truncated message,
 * candidates drawn until one is accepted (:151-158); no personalisation string (options.pers).
 * Arguments and outputs as ellgpu_ecdsa_sign without `nonces`; priv is n.byteLength() bytes,
 * big-endian, and -- like KeyPair#_importPrivate (ec/key.js:91-96) -- reduced mod n before it
 * seeds the DRBG, so a key >= n signs exactly like its residue.  out_ok[i] = 0 only
 * in the (never observed) case that the reference would go on to a further candidate because
 * r or s came out zero. */
int ellgpu_ecdsa_sign_det(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                          int msg_bits, const uint8_t* priv, int canonical, uint8_t* out_r, uint8_t* out_s,
                          uint8_t* out_recid, uint8_t* out_ok);
int ellgpu_ecdsa_sign_det_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                              int msg_bits, const uint8_t* priv, int canonical, uint8_t* out_r,
                              uint8_t* out_s, uint8_t* out_recid, uint8_t* out_ok, void* stream);

/* ECDSA public-key recovery: out_xy[i] = EC#recoverPubKey(hash_i, {r_i, s_i}, recid_i),
 * lib/elliptic/ec/index.js:231-259 -- Q = r^-1 (s R - e G) with R = pointFromX(r + (j >> 1) n,
 * j & 1) (short.js:187-204) and e = new BN(hash) (NOT truncated by that method, only reduced
 * mod n; hash_len bytes per item, at most twice the order width).  out_status[i]:
 *   0  Q is a finite point, written to out_xy (x || y)
 *   1  Q is the point at infinity
 *   2  the reference throws: 'Unable to find sencond key candinate' (j & 2 with r >= p mod n),
 *      'invalid point' (no point with that x), or j > 3
 *   3  outside the engine's domain (r = 0 or r >= n, where the reference inverts an unreduced
 *      BN): hand the item to the reference
 * out_xy is zeroed where the status is not 0.  Not available for p224 (p = 1 mod 4). */
int ellgpu_ecdsa_recover(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                         const uint8_t* r, const uint8_t* s, const uint8_t* recid, uint8_t* out_xy,
                         uint8_t* out_status);
int ellgpu_ecdsa_recover_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash, int hash_len,
                             const uint8_t* r, const uint8_t* s, const uint8_t* recid, uint8_t* out_xy,
                             uint8_t* out_status, void* stream);

/* EdDSA (ed25519) sign from 32-byte secrets: out_sig[i] = EDDSA#sign(msg_i, secret_i).toBytes()
 * (R || S, 64 bytes), lib/elliptic/eddsa/index.js:32-50, with the key derivation of
 * KeyPair.fromSecret (eddsa/key.js:42-75): hash = SHA-512(secret), a = the clamped first half,
 * prefix = the second half, A = a*G, r = SHA-512(prefix || M) mod n, R = r*G,
 * S = (r + SHA-512(R || A || M) * a) mod n.  Messages as for ellgpu_eddsa_verify.  out_pub (may be
 * NULL): the encoded public keys A (n x 32 bytes, KeyPair#getPublic). */
int ellgpu_eddsa_sign(ellgpu_ctx* ctx, size_t n, const uint8_t* secrets, const uint8_t* msgs,
                      const uint64_t* msg_off, size_t msg_len, uint8_t* out_sig, uint8_t* out_pub);
int ellgpu_eddsa_sign_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* secrets, const uint8_t* msgs,
                          const uint64_t* msg_off, size_t msg_len, uint8_t* out_sig, uint8_t* out_pub,
                          void* stream);

/* ---- device-buffer entry points (inputs/outputs resident in HBM) --------
 * Point operands and results must not overlap (in_xy / p1_xy / p2_xy against out_xy): the
 * operands are read again, for the curve test, after the results are written -- ELLGPU_E_ARG. */
int ellgpu_mul_fixed_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k,
                         uint8_t* out_xy, uint8_t* out_inf, void* stream);
int ellgpu_mul_var_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k,
                       const uint8_t* in_xy, uint8_t* out_xy, uint8_t* out_inf, void* stream);
int ellgpu_mul_add2_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* k1,
                        const uint8_t* p1_xy, const uint8_t* k2, const uint8_t* p2_xy,
                        uint8_t* out_xy, uint8_t* out_inf, void* stream);
int ellgpu_ecdsa_verify_dev(ellgpu_ctx* ctx, int curve, size_t n, const uint8_t* hash,
                            int hash_len, int msg_bits, const uint8_t* r, const uint8_t* s,
                            const uint8_t* pub_xy, uint8_t* out_ok, uint8_t* out_status,
                            void* stream);
int ellgpu_x25519_ladder_dev(ellgpu_ctx* ctx, size_t n, const uint8_t* k, const uint8_t* in_x,
                             uint8_t* out_x, uint8_t* out_inf, void* stream);

/* Pre-size the context's scratch arena for batches of up to n items of `curve`
 * (otherwise it grows on first use); also builds the curve's fixed-base table. */
int ellgpu_ctx_reserve(ellgpu_ctx* ctx, int curve, size_t n);
/* Window width, in bits, of the curve's fixed-base table on this context (0: not built yet; a
 * group answers for its first member).  The 256-bit curves' signed comb defaults to 22-bit
 * windows (12 additions per k*G, 1.6 GB); when the device cannot hold that table the engine
 * builds a narrower one -- 16, 12, 8 or 4 bits -- instead of failing with ELLGPU_E_NOMEM, and the
 * same kernels run on it (results are the same points; more additions per k*G).  The reference's
 * counterpart is the fixed `doubles` step of BasePoint#precompute (base.js:312-346). */
int ellgpu_ctx_comb_bits(ellgpu_ctx* ctx, int curve);

/* ---- measurement helper --------------------------------------------------
 * Integer-VALU roofline probe: runs `iters` dependent-free 32x32+64 -> 64 bit
 * multiply-accumulates (v_mad_u64_u32) per lane on `blocks` x 256 lanes and
 * returns the elapsed milliseconds measured with HIP events.  kind: 0 =
 * v_mad_u64_u32, 1 = v_mul_lo_u32 + v_mul_hi_u32 pair, 2 = v_mad_u32_u24,
 * 3 = v_add_co/addc pair.  Used by bench.py to set the VALU peak.
 * kind 10..15: secp256k1 field-layer probes, one wavefront per block of 64 lanes, `iters`
 * dependent operations per lane (blocks = 1024 * w puts w waves on every SIMD): 10 field mul,
 * 11 field sqr, 12 two interleaved mul chains, 13 add + sub, 14 Jacobian doubling,
 * 15 mixed addition, 16 wide product only, 17 wide square only, 18 reduction only;
 * 20 / 24 / 25: product / doubling / mixed addition of the lanes-per-item layer (one item per
 * wavefront: blocks items);
 * ops_out = lane-operations. */
int ellgpu_probe_valu(ellgpu_ctx* ctx, int kind, int blocks, int iters, double* ms_out,
                      double* ops_out);

/* White-box probe for the GPU unit tests of the field layer: r[i] = a[i] <op> b[i] in one of
 * the engine's fields (n items, `L` 32-bit little-endian limbs each, host pointers).
 * field: 0 secp256k1 p, 1 2^255-19, 10+c / 20+c base / order field of short curve c in
 * Montgomery form (26 = ed25519 order); op: 0 add 1 sub 2 mul 3 sqr 4 inv 5 neg, 6 / 7 / 8
 * multiply by 2 / 4 / 8, 9 identity (reduce the input), 10 (field 1 only) multiply by the
 * one-limb constant b[0]. */
int ellgpu_debug_field_op(ellgpu_ctx* ctx, int field, int op, size_t n, const uint32_t* a,
                          const uint32_t* b, uint32_t* r);

/* Per-kernel timing with HIP events recorded on the stream each kernel is
 * launched on.  set_timing(ctx, 1) starts a fresh recording, set_timing(ctx, 0)
 * stops it; get_timing synchronises the device and writes one text line per
 * kernel name, "<name> <launches> <total_ms>\n", returning the byte count. */
int ellgpu_ctx_set_timing(ellgpu_ctx* ctx, int on);
int ellgpu_ctx_get_timing(ellgpu_ctx* ctx, char* buf, size_t cap);

#ifdef __cplusplus
}
#endif
#endif /* ELLGPU_H */
