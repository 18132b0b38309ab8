'use strict';
// ellgpu -- JavaScript host layer: routes indutny/elliptic's scalar-multiplication
// hot path to libellgpu.so (MI355X) through the N-API addon, and adds the batch
// API the reference does not have.
//
//   var elliptic = require('elliptic');
//   var ellgpu = require('elliptic_amd/js').install(elliptic);      // patch in place
//   ec.verify(msg, sig, key)            // unchanged API, now one GPU launch per call
//   ellgpu.ecdsaVerifyBatch('secp256k1', {hashes, hashLen, r, s, pub})   // flat Buffers
//
// What install() replaces (reference file:line):
//   BaseCurve#_fixedNafMul      lib/elliptic/curve/base.js:52-84
//   BaseCurve#_wnafMul          lib/elliptic/curve/base.js:86-126
//   BaseCurve#_wnafMulAdd       lib/elliptic/curve/base.js:128-253   (len == 2)
//   ShortCurve#_endoWnafMulAdd  lib/elliptic/curve/short.js:218-249
//   mont Point#mul              lib/elliptic/curve/mont.js:130-153
// Contracts kept: same signatures; a NEW point object of the same class on the
// same curve is returned; infinity is curve.point(null, null) (short/mont) or the
// identity (edwards); jacobianResult=true returns an object with isInfinity() /
// eqXToP() (the affine result lifted with Z = 1); k is not reduced mod n; inputs
// are never mutated.  Anything outside the engine's domain -- a curve that is not
// one of the reference's presets (the toy curves of test/curve-test.js), a scalar
// wider than the curve's byte length, a negative scalar, a point that is NOT ON THE
// CURVE (the reference computes with those -- ec/index.js:192 never validates a key --
// and off the curve its answer depends on the order of its own operations; the engine
// reports such an item with status 2 instead of guessing) -- is handed to the
// reference's own original method, untouched.

var path = require('path');

var CURVES = ['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521', 'ed25519',
  'curve25519'];

function Engine(options) {
  options = options || {};
  this.addon = options.addon || require(options.addonPath ||
    path.join(__dirname, 'ellgpu.node'));
  this.addon.open(options.libPath || process.env.ELLGPU_LIB ||
    path.join(__dirname, '..', 'lib', 'libellgpu.so'));
  // options.devices = [d0, d1, ...]: a device GROUP -- mulBatch / mulAddBatch / ecdsaVerifyBatch
  // (and their Promise forms) shard every batch over the listed GPUs, one host thread per
  // device, results written straight into the result Buffers; everything else (and the
  // one-item calls of install()) runs on the first device.
  this.devices = Array.isArray(options.devices) && options.devices.length ?
    options.devices.map(function(d) { return d | 0; }) : null;
  this.ctx = this.addon.createContext(this.devices || (options.device | 0));
  this.stats = { gpuCalls: 0, gpuItems: 0, passthrough: 0, offCurve: 0 };
}

// The context owns the device's memory for this engine (fixed-base tables: 1.6 GB for secp256k1 at
// the default width) and is pinned by the addon until close(): call it when the engine is no
// longer needed (install(): eng.uninstall() first).  A process that simply ends needs no close().
Engine.prototype.close = function close() {
  if (!this.ctx) return;
  // (Promise-form batches are chained on the JS side before a worker thread gets them: _async)
  if (this._pending > 0) throw new Error('ellgpu: batches are in flight; close the engine when their Promises have settled');
  this.addon.destroyContext(this.ctx);
  this.ctx = null;
};
Engine.prototype._id = function _id(curve) {
  var id = typeof curve === 'number' ? curve : this.addon.curveId(curve);
  if (id < 0) throw new Error('Unknown curve ' + curve);
  return id;
};

// Window width of the curve's fixed-base table on this engine's (first) device: 0 before the first
// k*G, 22 on the 256-bit curves by default, 16 / 12 / 8 / 4 when the device could not hold that
// table (up to ~5x more additions per k*G: a crowded device shows up here, not only in the timings)
Engine.prototype.combBits = function combBits(curve) {
  return this.addon.combBits(this.ctx, this._id(curve));
};

// Register y^2 = x^3 + a x + b over an odd prime p < 2^256 that is none of the presets
// (`new elliptic.curve.short({p, a, b})`, lib/elliptic/curve/short.js:11-24) and return its curve
// id: usable with mulBatch (points given), mulAddBatch (both points given) and pointAddBatch, with
// 32-byte scalars and coordinates.  p, a, b: BN-like (toArray) or 32-byte Buffers.
Engine.prototype.defineShort = function defineShort(p, a, b) {
  function buf(v) {
    return Buffer.isBuffer(v) ? v : Buffer.from(v.toArray('be', 32));
  }
  return this.addon.defineShort(this.ctx, buf(p), buf(a), buf(b));
};
// The same for a (twisted) Edwards curve a x^2 + y^2 = 1 + d x^2 y^2 (c = 1) that is not ed25519
// (`new elliptic.curve.edwards({p, a, c: 1, d})`, lib/elliptic/curve/edwards.js:11-31)
Engine.prototype.defineEdwards = function defineEdwards(p, a, d) {
  function buf(v) {
    return Buffer.isBuffer(v) ? v : Buffer.from(v.toArray('be', 32));
  }
  return this.addon.defineEdwards(this.ctx, buf(p), buf(a), buf(d));
};

// ---- batch API on flat Buffers (fixed-width big-endian, item-major) ----------
// scalars: n x B bytes; points: n x 2B bytes (x||y) or null for the generator.
// -> { xy: Buffer(n x 2B), inf: Buffer(n) }.  out (optional): { xy, inf } Buffers of those sizes to
// write into -- a caller that keeps its result Buffers spares every call their allocation.
Engine.prototype.mulBatch = function mulBatch(curve, scalars, points, out) {
  var id = this._id(curve);
  var n = scalars.length / this.addon.fieldBytes(id);
  this.stats.gpuCalls++; this.stats.gpuItems += n;
  if (out)
    return points ? this.addon.mulVar(this.ctx, id, scalars, points, out.xy, out.inf) :
      this.addon.mulFixed(this.ctx, id, scalars, out.xy, out.inf);
  return points ? this.addon.mulVar(this.ctx, id, scalars, points) :
    this.addon.mulFixed(this.ctx, id, scalars);
};
// k1*P1 + k2*P2 per item; points1 == null means P1 = G
Engine.prototype.mulAddBatch = function mulAddBatch(curve, k1, points1, k2,
  points2) {
  var id = this._id(curve);
  this.stats.gpuCalls++;
  this.stats.gpuItems += k1.length / this.addon.fieldBytes(id);
  return this.addon.mulAdd2(this.ctx, id, k1, points1 || null, k2, points2);
};
// o = { hashes: Buffer(n x hashLen), hashLen, msgBits (0 = hashLen*8),
//       r: Buffer(n x NB), s: Buffer(n x NB), pub: Buffer(n x 2B) } -> Buffer(n), strictly 0 / 1:
// a mask.  o.status (optional Buffer(n)) receives the domain status per item: 2
// (Engine.OFF_CURVE) where r and s are in range but the key is not on the curve -- the verdict is
// 0 there, the safe answer; the reference computes with such keys and can answer true, and
// install() runs the reference on exactly those items -- else 0.
Engine.prototype.ecdsaVerifyBatch = function ecdsaVerifyBatch(curve, o) {
  var res = this.ecdsaVerifyRaw(curve, o);
  if (o.status) res.status.copy(o.status);
  return res.ok;
};
// the same, returning the addon's own result { ok: Buffer(n), status: Buffer(n) }.  The form a SPLIT call
// (addon.defer / addon.collect) must use: both Buffers are filled by collect(), so nothing may be read
// out of them -- and neither may be dropped -- before it
Engine.prototype.ecdsaVerifyRaw = function ecdsaVerifyRaw(curve, o) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += o.hashes.length / o.hashLen;
  return this.addon.ecdsaVerify(this.ctx, id, o.hashes, o.hashLen, o.msgBits | 0, o.r, o.s, o.pub);
};
Engine.prototype.x25519Batch = function x25519Batch(scalars, xs) {
  this.stats.gpuCalls++; this.stats.gpuItems += scalars.length / 32;
  return this.addon.x25519(this.ctx, scalars, xs);
};

// KeyPair#derive on curve25519 per item (ec/key.js:102-107): scalars, xs: Buffer(n x 32) ->
// { x: Buffer(n x 32), status: Buffer(n) }; status 0 shared secret, 1 x is no abscissa of the curve
// (the reference throws 'Assertion failed' out of its square root there), 2 the product is infinity
Engine.prototype.x25519DeriveBatch = function x25519DeriveBatch(scalars, xs) {
  this.stats.gpuCalls++; this.stats.gpuItems += scalars.length / 32;
  return this.addon.x25519Derive(this.ctx, scalars, xs);
};

// pointFromX (short curves: values = x) / pointFromY (ed25519: values = y), parity per item
// in `odd` (Buffer of 0/1) -> { xy: Buffer(n x 2B), ok: Buffer(n) }  (ok = 0: 'invalid point')
Engine.prototype.decompressBatch = function decompressBatch(curve, values, odd) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += odd.length;
  return this.addon.decompress(this.ctx, id, values, odd);
};

// ECDSA sign for supplied nonces (one pass of EC#sign's loop per item, ec/index.js:153-185).
// o = { hashes, hashLen, msgBits, priv: Buffer(n x NB), nonces: Buffer(n x NB), canonical }
// -> { r, s, recid, ok }   (ok = 0: the reference would try its next nonce)
Engine.prototype.ecdsaSignBatch = function ecdsaSignBatch(curve, o) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += o.hashes.length / o.hashLen;
  return this.addon.ecdsaSign(this.ctx, id, o.hashes, o.hashLen, o.msgBits | 0, o.priv, o.nonces,
    !!o.canonical);
};

// EC#sign with the reference's own deterministic nonces (HmacDRBG; ec/index.js:110-186).
// o = { hashes, hashLen, msgBits, priv: Buffer(n x NB), canonical } -> { r, s, recid, ok }
Engine.prototype.ecdsaSignDetBatch = function ecdsaSignDetBatch(curve, o) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += o.hashes.length / o.hashLen;
  return this.addon.ecdsaSignDet(this.ctx, id, o.hashes, o.hashLen, o.msgBits | 0, o.priv, !!o.canonical);
};

// EC#recoverPubKey per item (ec/index.js:231-259).  o = { hashes, hashLen, r, s: Buffer(n x NB),
// recid: Buffer(n) } -> { xy: Buffer(n x 2B), status: Buffer(n) }  (0 point, 1 infinity,
// 2 the reference throws, 3 outside the engine's domain: r = 0 or r >= n)
Engine.prototype.ecdsaRecoverBatch = function ecdsaRecoverBatch(curve, o) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += o.recid.length;
  return this.addon.ecdsaRecover(this.ctx, id, o.hashes, o.hashLen, o.r, o.s, o.recid);
};

// Point codecs and key validation on flat Buffers.
// decodePointBatch: enc = Buffer(n x encLen) of SEC1 encodings (02/03||x, 04/06/07||x||y) or, for
// ed25519, 32-byte EDDSA encodings -> { xy: Buffer(n x 2B), status: Buffer(n) }; status 0 point,
// 1 'Unknown point format', 2 'invalid point', 3 'Assertion failed' (hybrid prefix vs. y parity):
// the exception BaseCurve#decodePoint / EDDSA#decodePoint throws for that item.
Engine.prototype.decodePointBatch = function decodePointBatch(curve, enc, encLen) {
  var id = this._id(curve);
  this.stats.gpuCalls++; this.stats.gpuItems += enc.length / encLen;
  return this.addon.decodePoints(this.ctx, id, enc, encLen);
};
// encodePointBatch: xy = Buffer(n x 2B) -> Buffer(n x encLen) as BasePoint#encode(enc, compact)
// (EDDSA#encodePoint for ed25519, 32 bytes each)
Engine.prototype.encodePointBatch = function encodePointBatch(curve, xy, compact) {
  var id = this._id(curve);
  this.stats.gpuCalls++;
  return this.addon.encodePoints(this.ctx, id, xy, !!compact);
};
// validateBatch: KeyPair#validate per point -> Buffer(n) of 0 ok / 1 'Invalid public key'
// (o.inf[i] set) / 2 'Public key is not a point' / 3 'Public key * N != O' (skipped when
// o.checkOrder === false)
// status 2 in the `inf` result of mulBatch / mulAddBatch and in the `status` of ecdsaVerifyBatch
// (ELLGPU_STATUS_OFF_CURVE; ecdsaVerifyWireBatch: err 5): a point operand is not on the curve --
// outside the engine's domain, reported instead of guessed.  Verdicts (`ok`) stay 0 / 1.
Engine.OFF_CURVE = 2;
Engine.VALIDATE_REASON = [null, 'Invalid public key', 'Public key is not a point', 'Public key * N != O'];
Engine.prototype.validateBatch = function validateBatch(curve, xy, o) {
  var id = this._id(curve);
  o = o || {};
  this.stats.gpuCalls++;
  return this.addon.validate(this.ctx, id, xy, o.inf || null, o.checkOrder !== false);
};

// Point#add per pair of affine points: p, q Buffer(n x 2B); o.infP / o.infQ optional Buffer(n)
// flags for operands at infinity -> { xy, inf }
Engine.prototype.pointAddBatch = function pointAddBatch(curve, p, q, o) {
  o = o || {};
  this.stats.gpuCalls++;
  return this.addon.pointAdd(this.ctx, this._id(curve), p, o.infP || null, q, o.infQ || null);
};

// DER signatures and EC#verify on wire formats.
function packRecords(items) {
  var n = items.length, stride = 1, i;
  for (i = 0; i < n; i++) stride = Math.max(stride, items[i].length);
  var buf = Buffer.alloc(n * stride), lens = Buffer.alloc(n * 4);
  for (i = 0; i < n; i++) {
    Buffer.from(items[i]).copy(buf, i * stride);
    lens.writeUInt32LE(items[i].length, i * 4);
  }
  return { buf: buf, lens: lens, stride: stride };
}
// sigs: array of Buffers (DER) -> { r, s: Buffer(n x NB), status: Buffer(n) }; status 0 parsed,
// 1 where new Signature(der) throws 'Signature without r or s', 2 wider than the order
Engine.prototype.sigFromDerBatch = function sigFromDerBatch(curve, sigs) {
  var p = packRecords(sigs);
  this.stats.gpuCalls++;
  return this.addon.sigFromDer(this.ctx, this._id(curve), p.buf, p.stride, p.lens);
};
// r, s: Buffer(n x NB) -> array of Buffers, each Signature#toDER()
Engine.prototype.sigToDerBatch = function sigToDerBatch(curve, r, s) {
  this.stats.gpuCalls++;
  var o = this.addon.sigToDer(this.ctx, this._id(curve), r, s);
  var n = o.lens.length / 4, stride = n ? o.der.length / n : 0, out = [];
  for (var i = 0; i < n; i++) out.push(o.der.slice(i * stride, i * stride + o.lens.readUInt32LE(i * 4)));
  return out;
};
// EC#verify(msg, derSignature, encodedKey): hashes Buffer(n x hashLen), sigs array of DER Buffers,
// keys Buffer(n x keyLen) of SEC1 encodings -> { ok, err: Buffer(n) }; err 1..3 = decodePoint's
// exception for the key ('Unknown point format' / 'invalid point' / 'Assertion failed'),
// 4 = 'Signature without r or s'; ok is 0 wherever err is not 0
Engine.WIRE_ERROR = [null, 'Unknown point format', 'invalid point', 'Assertion failed', 'Signature without r or s',
  null /* 5: no exception -- the key is not on the curve: run the reference on this item if its answer is wanted */];
Engine.prototype.ecdsaVerifyWireBatch = function ecdsaVerifyWireBatch(curve, o) {
  var p = packRecords(o.sigs);
  this.stats.gpuCalls++; this.stats.gpuItems += o.sigs.length;
  return this.addon.ecdsaVerifyWire(this.ctx, this._id(curve), o.hashes, o.hashLen, o.msgBits | 0,
    p.buf, p.stride, p.lens, o.keys, o.keyLen);
};

// KeyPair#derive per item (ec/key.js:102-107): priv Buffer(n x B), pub Buffer(n x 2B) ->
// { x: Buffer(n x B), status: Buffer(n) }; status 0 shared secret, 1 'public point not validated',
// 2 the product is the point at infinity (the reference's getX throws)
Engine.prototype.ecdhDeriveBatch = function ecdhDeriveBatch(curve, priv, pub) {
  var st = this.validateBatch(curve, pub, { checkOrder: false });
  var r = this.mulBatch(curve, priv, pub);
  var n = st.length, B = n ? r.xy.length / n / 2 : 0;
  var x = Buffer.alloc(n * B), status = Buffer.alloc(n);
  for (var i = 0; i < n; i++) {
    status[i] = st[i] ? 1 : (r.inf[i] ? 2 : 0);
    if (!status[i]) r.xy.copy(x, i * B, i * 2 * B, i * 2 * B + B);
  }
  return { x: x, status: status };
};

// ed25519 EdDSA verify.  msgs: array of Buffers (any lengths); sigs: Buffer(n x 64) of R||S;
// pubs: Buffer(n x 32).  -> { ok: Buffer(n), err: Buffer(n) }  (err = 1 where the
// reference throws: R or A is not a curve point)
Engine.prototype.eddsaVerifyBatch = function eddsaVerifyBatch(msgs, sigs, pubs) {
  var n = msgs.length;
  var off = Buffer.alloc((n + 1) * 8);
  var pos = 0;
  for (var i = 0; i < n; i++) {
    off.writeUInt32LE(pos >>> 0, i * 8); off.writeUInt32LE(Math.floor(pos / 4294967296), i * 8 + 4);
    pos += msgs[i].length;
  }
  off.writeUInt32LE(pos >>> 0, n * 8); off.writeUInt32LE(Math.floor(pos / 4294967296), n * 8 + 4);
  this.stats.gpuCalls++; this.stats.gpuItems += n;
  return this.addon.eddsaVerify(this.ctx, Buffer.concat(msgs), off, 0, sigs, pubs);
};

// ed25519 EdDSA sign from 32-byte secrets (EDDSA#sign with keyFromSecret).  msgs: array of
// Buffers; secrets: Buffer(n x 32).  -> { sig: Buffer(n x 64) of R||S, pub: Buffer(n x 32) }
Engine.prototype.eddsaSignBatch = function eddsaSignBatch(msgs, secrets) {
  var n = msgs.length;
  var off = Buffer.alloc((n + 1) * 8);
  var pos = 0;
  for (var i = 0; i < n; i++) {
    off.writeUInt32LE(pos >>> 0, i * 8); off.writeUInt32LE(Math.floor(pos / 4294967296), i * 8 + 4);
    pos += msgs[i].length;
  }
  off.writeUInt32LE(pos >>> 0, n * 8); off.writeUInt32LE(Math.floor(pos / 4294967296), n * 8 + 4);
  this.stats.gpuCalls++; this.stats.gpuItems += n;
  return this.addon.eddsaSign(this.ctx, Buffer.concat(msgs), off, 0, secrets);
};

// ---- asynchronous batch API: same arguments, returns a Promise; the work runs on a
// libuv worker thread (napi_async_work), so the JS thread stays responsive during a large
// batch.  A context processes one call at a time, so calls are chained.
Engine.prototype._async = function _async(op, curve, hashLen, msgBits, b0, b1, b2, b3, i0, i1) {
  var self = this;
  var id = op === 4 ? 7 : this._id(curve);
  var run = function() {
    self.stats.gpuCalls++;
    return self.addon.callAsync(op, self.ctx, id, hashLen | 0, msgBits | 0, b0 || null,
      b1 || null, b2 || null, b3 || null, i0 | 0, i1 | 0);
  };
  var p = (this._tail || Promise.resolve()).then(run, run);
  this._tail = p.catch(function() {});
  this._pending = (this._pending | 0) + 1;
  var settled = function() { self._pending--; };
  p.then(settled, settled);
  return p;
};
Engine.prototype.mulBatchAsync = function(curve, scalars, points) {
  return points ? this._async(1, curve, 0, 0, scalars, points) : this._async(0, curve, 0, 0, scalars);
};
Engine.prototype.mulAddBatchAsync = function(curve, k1, points1, k2, points2) {
  return this._async(2, curve, 0, 0, k1, points1 || null, k2, points2);
};
Engine.prototype.ecdsaVerifyBatchAsync = function(curve, o) {
  return this._async(3, curve, o.hashLen, o.msgBits | 0, o.hashes, o.r, o.s, o.pub).then(function(res) {
    if (o.status) res.status.copy(o.status);
    return res.ok;
  });
};
Engine.prototype.ecdsaSignDetBatchAsync = function(curve, o) {
  return this._async(5, curve, o.hashLen, o.msgBits | 0, o.hashes, o.priv, null, null, o.canonical ? 1 : 0, 0);
};
Engine.prototype.ecdsaRecoverBatchAsync = function(curve, o) {
  return this._async(6, curve, o.hashLen, 0, o.hashes, o.r, o.s, o.recid);
};
Engine.prototype.ecdsaVerifyWireBatchAsync = function(curve, o) {
  var p = packRecords(o.sigs);
  return this._async(7, curve, o.hashLen, o.msgBits | 0, o.hashes, p.buf, p.lens, o.keys, p.stride, o.keyLen);
};
Engine.prototype.decodePointBatchAsync = function(curve, enc, encLen) {
  return this._async(8, curve, 0, 0, enc, null, null, null, encLen, 0);
};
Engine.prototype.x25519BatchAsync = function(scalars, xs) {
  return this._async(4, 7, 0, 0, scalars, xs);
};

// ---- install(): prototype patch on a user-supplied elliptic instance ----------
function install(elliptic, options) {
  var eng = new Engine(options);
  var BN = elliptic.curves.secp256k1.curve.p.constructor;
  var addon = eng.addon;

  // preset lookup: a curve object is in the engine's domain iff its type, p, coefficients,
  // generator AND order equal one of the reference's presets (the engine's fixed-base tables,
  // GLV constants and scalar field belong to the preset's G and n: a curve over a preset field
  // with another generator -- or without one -- is passed through to the reference's own code)
  // (the presets' numbers are COPIED here, when install() runs: the library's own preset objects are as
  // reachable to the caller as any other curve object -- `ec.curve` of `new EC('secp256k1')` IS
  // elliptic.curves.secp256k1.curve -- and a preset edited afterwards must stop being recognised)
  // hash.js's sha512 as the EDDSA constructor assigns it to every instance (eddsa/index.js:24): taken
  // from a fresh instance made at the first patched EDDSA call -- by then the caller's own instance
  // exists, so the constructor's one side effect (G's precompute) has already happened
  var HASH512 = null;
  var presets = {};
  CURVES.forEach(function(name) {
    var c = elliptic.curves[name].curve;
    var g = genCoords(c, c.g);
    presets[c.type + ':' + c.p.toString(16)] = { name: name, id: addon.curveId(name),
      B: addon.fieldBytes(addon.curveId(name)), g: c.g, hash: elliptic.curves[name].hash,
      ref: { a: c.a.fromRed().clone(), b: c.b && c.b.red ? c.b.fromRed().clone() : null,
        d: c.d && c.d.red ? c.d.fromRed().clone() : null, n: c.n.clone(),
        g: g && g.map(function(v) { return v.clone(); }) } };
  });
  // affine coordinates of a generator as plain BNs, without touching the point object
  function genCoords(curve, g) {
    if (curve.type === 'short') return g.inf ? null : [g.x.fromRed(), g.y.fromRed()];
    if (curve.type === 'edwards') {
      var q = curve.point(g.x, g.y, g.z, g.t);        // clone: getX() normalizes in place
      return [q.getX(), q.getY()];
    }
    var m = curve.point(g.x, g.z);
    return m.isInfinity() ? null : [m.getX()];
  }
  // (ref: the numbers install() copied from the preset)
  function sameGenerator(curve, ref) {
    if (!curve.g || !curve.n || !BN.isBN(curve.n) || curve.n.red || curve.n.cmp(ref.n) !== 0) return false;
    var a = genCoords(curve, curve.g), b = ref.g;
    if (!a || !b || a.length !== b.length) return false;
    for (var i = 0; i < a.length; i++) if (a[i].cmp(b[i]) !== 0) return false;
    return true;
  }
  // What is remembered is kept in WeakMaps of this install(), keyed by the caller's object: nothing is
  // written onto the caller's objects (frozen ones included), and nobody else can plant a verdict.
  var memo = { _ellgpu: new WeakMap(), _ellgpuCustom: new WeakMap(), _ellgpuEndo: new WeakMap(), _ellgpuOK: new WeakMap() };
  function hide(o, k, v) {
    if (o !== null && (typeof o === 'object' || typeof o === 'function')) memo[k].set(o, v);
  }
  function recall(o, k) {
    return o !== null && (typeof o === 'object' || typeof o === 'function') ? memo[k].get(o) : undefined;
  }
  // ---- witnesses: what a remembered verdict stands on ---------------------------------------------
  // install() remembers what it has established about the caller's objects -- "this curve object is
  // the preset", "these are the curve's GLV constants", "this table holds its point's multiples" --
  // because establishing it costs up to tens of milliseconds.  But the objects stay the caller's, and
  // the reference reads them afresh on every call: a table entry replaced in place, an entry's
  // coordinate edited, `curve.n` or a basis vector changed after the first call change the
  // reference's answer and must change the patched library's.  A remembered verdict therefore carries
  // a WITNESS -- a flat copy of every number it depended on -- and is used only while the live objects
  // still hold exactly those numbers (an exact comparison, not a hash); otherwise the check is made
  // again.  Re-reading a secp256k1 G's 388 table entries costs ~10 us of host time; the one-item calls
  // hide it behind the device's work (`guarded`, below).  What is NOT re-read is stated in
  // INTEGRATION.md section 2 ("what install() treats as immutable").
  // a BN as [negative, length, words...]; anything that is no BN as a marker no BN compares equal to
  function snapBN(v, b) {
    if (!b || typeof b !== 'object' || !Array.isArray(b.words) || !(b.length >= 0 && b.length <= b.words.length)) { v.push(-2); return; }
    v.push(b.negative | 0, b.length);
    for (var i = 0; i < b.length; i++) v.push(b.words[i] | 0);
  }
  // -> position behind the BN in the snapshot, or -1: another value, or not in reduction context `red`
  function sameBN(s, pos, b, red) {
    if (s[pos] === -2) return b === undefined || b === null ? pos + 1 : -1;
    if (!b || b.red !== red || b.negative !== s[pos]) return -1;
    var l = b.length, w = b.words;
    if (s[pos + 1] !== l || !w) return -1;
    pos += 2;
    for (var j = 0; j < l; j++) if (w[j] !== s[pos + j]) return -1;
    return pos + l;
  }
  // a table entry / an operand with the fields the reference's formulas read (short.js:365-412,
  // 569-603: x, y, inf; edwards.js:174-348: x, y, z, t, zOne)
  function snapEntry(curve, v, e) {
    if (!e || typeof e !== 'object') { v.push(-3); return; }
    if (curve.type === 'short') { snapBN(v, e.x); snapBN(v, e.y); return; }
    v.push(e.zOne ? 1 : 0);
    snapBN(v, e.x); snapBN(v, e.y); snapBN(v, e.z);
    if (curve.extended) snapBN(v, e.t);
  }
  function sameEntry(curve, s, pos, e) {
    if (s[pos] === -3) return e === undefined || e === null ? pos + 1 : -1;
    if (!e || e.curve !== curve) return -1;
    var red = curve.red;
    if (curve.type === 'short') {
      if (e.inf !== false || e.type !== 'affine') return -1;
      pos = sameBN(s, pos, e.x, red);
      return pos < 0 ? -1 : sameBN(s, pos, e.y, red);
    }
    if (s[pos] !== (e.zOne ? 1 : 0)) return -1;
    pos = sameBN(s, pos + 1, e.x, red);
    if (pos >= 0) pos = sameBN(s, pos, e.y, red);
    if (pos >= 0) pos = sameBN(s, pos, e.z, red);
    if (pos >= 0 && curve.extended) pos = sameBN(s, pos, e.t, red);
    return pos;
  }
  // the numbers a curve object is recognised by: p, its coefficients, n and the generator
  function coeffs(curve) {
    return curve.type === 'edwards' ? [curve.a, curve.d, curve.c] : [curve.a, curve.b];
  }
  // ... and what the constructors DERIVE from them and the formulas read (base.js:20-23 zero / one /
  // two, short.js:20-25 tinv / zeroA / threeA, edwards.js:15-32 c2 / dd / oneC / twisted / mOneA /
  // extended, mont.js:14-19 i4 / a24): part of every witness, and right -- derivedOK -- or the curve
  // object is the reference's own
  function derived(curve) {
    if (curve.type === 'short') return [curve.zero, curve.one, curve.two, curve.tinv];
    if (curve.type === 'edwards') return [curve.zero, curve.one, curve.two, curve.c2, curve.dd];
    return [curve.zero, curve.one, curve.two, curve.i4, curve.a24];
  }
  function flagsOf(curve) {
    // (bit 4: getNAF(k, w, curve._bitLength), base.js:56, 96, sizes an array with it -- any integer will do)
    var b = Number.isInteger(curve._bitLength) ? 16 : 0;
    if (curve.type === 'short') return b | (curve.zeroA ? 1 : 0) | (curve.threeA ? 2 : 0);
    if (curve.type === 'edwards') return b | (curve.twisted ? 1 : 0) | (curve.mOneA ? 2 : 0) | (curve.extended ? 4 : 0) | (curve.oneC ? 8 : 0);
    return b;
  }
  function derivedOK(curve) {
    var red = curve.red, one = new BN(1).toRed(red), two = one.redAdd(one);
    function is(v, w) { return !!v && v.red === red && v.negative === 0 && v.cmp(w) === 0; }
    if (!is(curve.zero, new BN(0).toRed(red)) || !is(curve.one, one) || !is(curve.two, two)) return false;
    if (curve.type !== 'mont' && !Number.isInteger(curve._bitLength)) return false;
    var a = curve.a, m1 = one.redNeg();
    if (!a || a.red !== red || a.negative !== 0) return false;
    if (curve.type === 'short') {
      return is(curve.tinv, two.redInvm()) && !!curve.zeroA === (a.cmpn(0) === 0) &&
        !!curve.threeA === (a.cmp(m1.redSub(two)) === 0);
    }
    if (curve.type === 'edwards') {
      var c = curve.c, d = curve.d;
      if (!c || c.red !== red || !d || d.red !== red || c.negative !== 0 || d.negative !== 0) return false;
      return is(curve.c2, c.redSqr()) && is(curve.dd, d.redAdd(d)) && !!curve.oneC === (c.cmp(one) === 0) &&
        !!curve.extended === !!curve.mOneA && (!curve.mOneA || (!!curve.twisted && a.cmp(m1) === 0)) && (!!curve.twisted || a.cmp(one) === 0);
    }
    var i4 = two.redAdd(two).redInvm();
    return is(curve.i4, i4) && is(curve.a24, i4.redMul(a.redAdd(two)));
  }
  function snapCurve(curve, withGen) {
    var v = [];
    snapBN(v, curve.p);
    coeffs(curve).forEach(function(c) { snapBN(v, c); });
    derived(curve).forEach(function(c) { snapBN(v, c); });
    v.push(flagsOf(curve));
    if (withGen) {
      snapBN(v, curve.n);
      var g = curve.g;
      if (!g || typeof g !== 'object') v.push(-3);
      else if (curve.type === 'mont') { snapBN(v, g.x); snapBN(v, g.z); }
      else snapEntry(curve, v, g);
    }
    return Int32Array.from(v);
  }
  function sameCurve(curve, s, withGen) {
    var red = curve.red, pos = sameBN(s, 0, curve.p, null), cs = coeffs(curve);
    for (var i = 0; pos >= 0 && i < cs.length; i++) pos = sameBN(s, pos, cs[i], red);
    var ds = derived(curve);
    for (i = 0; pos >= 0 && i < ds.length; i++) pos = sameBN(s, pos, ds[i], red);
    if (pos >= 0) pos = s[pos] === flagsOf(curve) ? pos + 1 : -1;
    if (pos >= 0 && withGen) {
      pos = sameBN(s, pos, curve.n, null);
      var g = curve.g;
      if (pos >= 0 && !g) pos = -1;
      else if (pos >= 0 && curve.type === 'mont') { pos = sameBN(s, pos, g.x, red); if (pos >= 0) pos = sameBN(s, pos, g.z, red); }
      else if (pos >= 0) pos = sameEntry(curve, s, pos, g);
    }
    return pos === s.length;
  }
  // (whatever the caller's objects make these checks throw -- a coefficient that is no BN any more, a
  // sign bit set on a reduced number -- is an object the patch cannot vouch for: the reference's own code)
  function domain(curve) {
    try { return domain0(curve); } catch (e) { return null; }
  }
  function domain0(curve) {
    var c = recall(curve, '_ellgpu');
    if (c !== undefined && c.red === curve.red && sameCurve(curve, c.snap, true)) return c.d;
    var d = presets[curve.type + ':' + curve.p.toString(16)] || null;
    // (methods of a curve under construction -- ShortCurve#_getEndomorphism multiplies g before
    // the constructor returns -- see every field this test reads: p, n, g, a, b are set first)
    if (d && curve.type === 'short') {
      if (!curve.a || !curve.b || curve.a.fromRed().cmp(d.ref.a) !== 0 ||
          curve.b.fromRed().cmp(d.ref.b) !== 0) d = null;
    } else if (d && curve.type === 'edwards') {
      if (curve.a.fromRed().cmp(d.ref.a) !== 0 ||
          curve.d.fromRed().cmp(d.ref.d) !== 0 || !curve.extended) d = null;
    } else if (d && curve.type === 'mont') {
      if (curve.a.fromRed().cmp(d.ref.a) !== 0 || !curve.b || curve.b.fromRed().cmp(d.ref.b) !== 0) d = null;
    }
    if (d && !sameGenerator(curve, d.ref)) d = null;
    if (d && !derivedOK(curve)) d = null;
    var snap = null;
    try { snap = snapCurve(curve, true); } catch (e) { snap = null; }
    if (snap) hide(curve, '_ellgpu', { d: d, red: curve.red, snap: snap });
    return d;
  }
  // A short curve that is no preset still gets its Point#mul / mulAdd / jmulAdd from the device:
  // run-time prime (<= 256 bits), arbitrary a (the generic `_dbl` / `dblp` of short.js:802-830,
  // 605-654), no fixed-base table.  options.customCurves === false keeps such curves on the
  // reference's own code, as do primes wider than 256 bits and a seventeenth distinct curve.
  // Miller-Rabin to twelve prime bases and eight RANDOM ones (composites that pass any fixed set of
  // bases can be constructed -- Arnault 1995 -- but not ones that pass bases drawn after the number
  // was chosen: error < 4^-8 per call, for adversarial input only): the device inverts by Fermat's
  // a^(p-2), which equals the reference's extended-Euclid BN#invm only for a prime modulus -- a
  // composite `p` (which the reference accepts) stays on the reference's own code
  function probablyPrime(p) {
    var small = [2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37];
    for (var i = 0; i < small.length; i++) {
      if (p.cmpn(small[i]) === 0) return true;
      if (p.modn(small[i]) === 0) return false;
    }
    var red = BN.red(p);
    var pm1 = p.subn(1);
    var s = 0;
    var dd = pm1.clone();
    while (dd.isEven()) { dd = dd.shrn(1); s++; }
    var one = new BN(1).toRed(red);
    var m1 = pm1.toRed(red);
    var bases = small.map(function(b) { return new BN(b); });
    if (p.bitLength() > 48) {
      var rb = require('crypto').randomBytes(8 * p.byteLength());
      for (i = 0; i < 8; i++) {
        var a = new BN(rb.slice(i * p.byteLength(), (i + 1) * p.byteLength())).umod(p.subn(3)).iaddn(2);   // [2, p - 2]
        bases.push(a);
      }
    }
    for (i = 0; i < bases.length; i++) {
      var x = bases[i].toRed(red).redPow(dd);
      if (x.cmp(one) === 0 || x.cmp(m1) === 0) continue;
      var composite = true;
      for (var r = 1; r < s; r++) {
        x = x.redSqr();
        if (x.cmp(m1) === 0) { composite = false; break; }
      }
      if (composite) return false;
    }
    return true;
  }
  function customDomain(curve) {
    try { return customDomain0(curve); } catch (e) { return null; }
  }
  function customDomain0(curve) {
    var cc = recall(curve, '_ellgpuCustom');
    if (cc !== undefined && cc.red === curve.red && sameCurve(curve, cc.snap, false)) return cc.d;
    var d = null;
    if (options && options.customCurves === false) return null;
    function remember(v) {
      var snap = null;
      try { snap = snapCurve(curve, false); } catch (e) { snap = null; }
      if (snap) hide(curve, '_ellgpuCustom', { d: v, red: curve.red, snap: snap });
      return v;
    }
    if (curve.p && curve.p.bitLength() <= 256 && curve.p.isOdd() && curve.p.cmpn(3) > 0 &&
        !probablyPrime(curve.p)) return remember(null);
    if (!derivedOK(curve)) return remember(null);
    // A singular cubic (4 a^3 + 27 b^2 = 0) has no group law at its singular point, and an Edwards
    // curve whose addition law is not complete (complete: a a square, d not -- Bernstein et al.,
    // "Twisted Edwards curves", section 6) has pairs of points on which the projective formulas give
    // Z = 0: there the reference's result depends on the order in which ITS ladder adds, and is not
    // the engine's.  Such curves (exhaustive search over F_5, F_7, F_11: the only disagreements) stay
    // on the reference's own code.
    var red = curve.red, fe = function(v) { return new BN(v).toRed(red); };
    if (curve.type === 'short' && curve.a && curve.b && curve.p.bitLength() <= 256 &&
        curve.p.isOdd() && curve.p.cmpn(3) > 0 &&
        curve.a.redSqr().redMul(curve.a).redMul(fe(4)).redAdd(curve.b.redSqr().redMul(fe(27))).cmpn(0) !== 0) {
      try {
        d = { name: 'custom', custom: true, B: 32,
          id: eng.defineShort(curve.p, curve.a.fromRed(), curve.b.fromRed()) };
      } catch (e) { d = null; }
    } else if (curve.type === 'edwards' && curve.a && curve.d && curve.c &&
        curve.c.fromRed().cmpn(1) === 0 && curve.p.bitLength() <= 256 && curve.p.isOdd() &&
        curve.p.cmpn(3) > 0 && curve.a.cmp(curve.d) !== 0 &&
        curve.a.redPow(curve.p.subn(1).ushrn(1)).cmp(fe(1)) === 0 &&
        curve.d.redPow(curve.p.subn(1).ushrn(1)).cmp(fe(1).redNeg()) === 0) {
      // (twisted) Edwards curves other than ed25519, c = 1: projective ladder on the device
      // (the reference's _projDbl / _projAdd for a != -1, its extended forms for a = -1)
      try {
        d = { name: 'custom-edwards', custom: true, B: 32,
          id: eng.defineEdwards(curve.p, curve.a.fromRed(), curve.d.fromRed()) };
      } catch (e) { d = null; }
    }
    return remember(d);
  }
  function scalarBuf(k, B) {
    if (!BN.isBN(k) || k.isNeg() || k.byteLength() > B) return null;
    return Buffer.from(k.toArray('be', B));
  }
  // affine (x, y) of a point without mutating it; null for infinity
  function affineBuf(curve, p, B) {
    // (a point object with a coordinate missing, or a number whose bn.js invariants are broken: whatever
    // looking at it throws, the reference's own method throws its own way)
    try { return affineBuf0(curve, p, B); } catch (e) { return null; }
  }
  function affineBuf0(curve, p, B) {
    // (a point of ANOTHER curve object -- the public half of a KeyPair made by another EC instance:
    // the reference's field operations throw 'red works only with red numbers' on it, its own to throw)
    if (!p || p.curve !== curve || p.isInfinity()) return null;
    var x, y;
    // (the reference's _wnafMulAdd also takes Jacobian points, base.js:158-183, 222-233: its own)
    // (coordinates that live in ANOTHER reduction context -- curve.point(Q.x, Q.y) with Q from another
    // curve object over the same field, which is also what ec.keyFromPublic(Q) does, ec/key.js:96 --
    // make bn.js throw 'red works only with red numbers' in the reference's first field operation)
    if (!p.x || !p.y || p.x.red !== curve.red || p.y.red !== curve.red) return null;
    // (a reduced number with its sign bit set: bn.js throws 'red works only with positives' at the first product)
    if (p.x.negative !== 0 || p.y.negative !== 0) return null;
    if (curve.type === 'short') { if (p.type !== 'affine') return null; x = p.getX(); y = p.getY(); }
    else {
      if (!p.z || p.z.red !== curve.red || (p.t && p.t.red !== curve.red)) return null;
      if (p.z.negative !== 0 || (p.t && p.t.negative !== 0)) return null;
      // extended coordinates carry T = X Y / Z, which the reference's _extAdd / _extDbl USE
      // (edwards.js:279-309): a point built with any other T (curve.point(x, y, z, t) takes what
      // it is given) is not the point its (x, y) says -- the reference's own, like a point that is
      // off the curve
      // (curves with a != -1 use the projective formulas, which never look at T)
      if (curve.extended && (!p.t || !p.x.red || !p.t.red || p.t.redMul(p.z).cmp(p.x.redMul(p.y)) !== 0)) return null;
      var q = curve.point(p.x, p.y, p.z, p.t);        // clone: getX() normalizes in place
      x = q.getX(); y = q.getY();
    }
    return Buffer.concat([Buffer.from(x.toArray('be', B)),
      Buffer.from(y.toArray('be', B))]);
  }
  function isG(curve, d, p) {
    if (d.custom) return false;
    if (curve.type === 'short')
      return !p.inf && p.x.cmp(curve.g.x) === 0 && p.y.cmp(curve.g.y) === 0;
    return p === curve.g;
  }
  // ---- inputs the reference TRUSTS: precomputed tables and the endomorphism's constants --------
  // A point's `precomputed` tables are input to the reference's ladders: _fixedNafMul adds
  // doubles.points[j] (base.js:52-84), _wnafMul / _wnafMulAdd add naf.points[(z - 1) >> 1]
  // (base.js:86-253, through _getNAFPoints, base.js:357-374), _endoWnafMulAdd takes
  // precomputed.beta for lambda * P (short.js:282-310) -- whatever they hold.  precompute() fills
  // them with the true multiples, but curve.pointFromJSON([x, y, { doubles, naf }])
  // (short.js:328-355, edwards.js the same way) takes them from the caller, and a table that is
  // not the point's multiples makes the reference return something else than k * P.  The engine
  // computes k * P from (x, y) alone, so it answers only for points whose tables ARE their
  // multiples: each table is checked once (the reference's Jacobian / projective dbl() and add(), a
  // few milliseconds per curve -- 6 for secp256k1, 65 for p521 --), and the verdict
  // is remembered FOR the table object (a WeakMap of this install(), beside a witness of what it was
  // reached on: `recall` / `hide`); anything else is left to the reference's own ladders.
  function canonical(curve, v) { return !!v && v.red === curve.red && !v.isNeg() && v.cmp(curve.p) < 0; }
  function entryOK(curve, e) {
    if (!e || e.curve !== curve || !canonical(curve, e.x) || !canonical(curve, e.y)) return false;
    if (curve.type === 'short') return e.type === 'affine' && e.inf === false;
    if (!canonical(curve, e.z) || e.z.cmpn(0) === 0) return false;
    if (e.zOne && e.z.cmp(curve.one) !== 0) return false;
    // extended coordinates: _extAdd multiplies the entries' T (edwards.js:279-309)
    if (curve.extended && (!canonical(curve, e.t) || e.t.redMul(e.z).cmp(e.x.redMul(e.y)) !== 0)) return false;
    return true;
  }
  // a table entry against the multiple it should be (q: a Jacobian point on a short curve -- no
  // inversions -- a projective / extended one on an Edwards curve): cross-multiplied coordinates
  function samePoint(curve, e, q) {
    if (!entryOK(curve, e) || !q) return false;
    if (curve.type === 'short') return !q.isInfinity() && q.eq(e);
    return e.x.redMul(q.z).cmp(q.x.redMul(e.z)) === 0 && e.y.redMul(q.z).cmp(q.y.redMul(e.z)) === 0;
  }
  function checkTable(curve, p, tbl, kind) {
    var pts = tbl.points, w = kind === 'naf' ? tbl.wnd : tbl.step;
    if (!Array.isArray(pts) || pts.length > 4096 || !Number.isInteger(w) || w < 1 || w > 12 || !entryOK(curve, p)) return false;
    var i, q = curve.type === 'short' ? p.toJ() : p;
    if (!samePoint(curve, pts[0], q)) return false;
    if (kind === 'naf') {
      // entry i is (2 i + 1) * P; a digit of getNAF(k, wnd) reaches index 2^(wnd-1) - 1 at most
      var need = 1 << (w - 1);
      if (pts.length < need) return false;
      var two = need > 1 ? q.dbl() : null;
      for (i = 1; i < need; i++) {
        q = q.add(two);
        if (!samePoint(curve, pts[i], q)) return false;
      }
      // (entries no digit reaches are never ADDED, but _getBeta and neg(true) map the whole array,
      // short.js:282-310, 437-462: each must be something those maps do not throw on)
      for (; i < pts.length; i++) if (!entryOK(curve, pts[i])) return false;
      return true;
    }
    // doubles: entry j is 2^(step j) * P, every one of them within reach of _hasDoubles
    for (i = 1; i < pts.length; i++) {
      for (var s = 0; s < w; s++) q = q.dbl();
      if (!samePoint(curve, pts[i], q)) return false;
    }
    return true;
  }
  // the table's witness: the point itself and every entry of `points` (all of them: _getBeta and
  // neg(true), short.js:282-310, 437-462, map the whole array)
  function snapTable(curve, p, pts) {
    var v = [];
    snapEntry(curve, v, p);
    for (var i = 0; i < pts.length; i++) snapEntry(curve, v, pts[i]);
    return Int32Array.from(v);
  }
  function sameTable(curve, p, pts, c) {
    if (pts.length !== c.len) return false;
    var s = c.snap, pos = sameEntry(curve, s, 0, p);
    for (var i = 0, n = c.len; pos >= 0 && i < n; i++) pos = sameEntry(curve, s, pos, pts[i]);
    return pos === s.length;
  }
  function tableOK(curve, p, tbl, kind) {
    if (!tbl) return true;
    var c = recall(tbl, '_ellgpuOK');
    var w = kind === 'naf' ? tbl.wnd : tbl.step;
    if (c && c.p === p && c.pts === tbl.points && c.w === w && sameTable(curve, p, c.pts, c)) return c.ok;
    var ok = false;
    refOnly++;
    try { ok = typeof tbl === 'object' && checkTable(curve, p, tbl, kind); } catch (e) { ok = false; } finally { refOnly--; }
    // (a table that is not even an array of a sane length is looked at again next time: that is cheap)
    if (typeof tbl === 'object' && Array.isArray(tbl.points) && tbl.points.length <= 4096) {
      var snap = null;
      try { snap = snapTable(curve, p, tbl.points); } catch (e) { snap = null; }
      if (snap) hide(tbl, '_ellgpuOK', { p: p, pts: tbl.points, len: tbl.points.length, w: w, ok: ok, snap: snap });
    }
    return ok;
  }
  function tablesOK(curve, p, inner) {
    try { return tablesOK0(curve, p, inner); } catch (e) { return false; }
  }
  function tablesOK0(curve, p, inner) {
    var pre = p && p.precomputed;
    if (!pre) return true;
    if (typeof pre !== 'object') return false;
    if (!tableOK(curve, p, pre.naf, 'naf') || !tableOK(curve, p, pre.doubles, 'doubles')) return false;
    if (pre.beta && curve.type === 'short' && curve.endo) {
      // lambda * P as _getBeta caches it: (beta x, y), with the tables of THAT point
      var b = pre.beta;
      if (inner || !entryOK(curve, b) || b.x.cmp(p.x.redMul(curve.endo.beta)) !== 0 || b.y.cmp(p.y) !== 0) return false;
      if (!tablesOK0(curve, b, true)) return false;
    }
    return true;
  }
  // The GLV ladder of the reference (short.js:168-249) is k * P only if its constants are what
  // _getEndomorphism would compute: conf.beta / conf.lambda / conf.basis are taken as given
  // (short.js:36-67), and k1 + k2 lambda = k (mod n) says nothing about a point outside <G> -- on a
  // curve whose group is larger than n the reference's result for such a point is not k * P.  The
  // engine answers for a curve with an endomorphism only when it can see that the two agree on
  // EVERY point of the curve: beta^3 = 1, n prime with n * G = O and lambda * G = (beta x_G, y_G), n
  // the only multiple of itself in the Hasse interval (so the group IS <G>), both basis vectors in
  // the lattice a + b lambda = 0 (mod n).  Checked once per curve object (two multiplications by
  // the reference's plain w-NAF ladder).
  function checkEndo(curve) {
    var e = curve.endo, n = curve.n, g = curve.g, p = curve.p;
    if (!e.beta || e.beta.red !== curve.red || !BN.isBN(e.lambda) || e.lambda.red || !Array.isArray(e.basis) || e.basis.length !== 2) return false;
    if (!n || !g || g.inf !== false || n.isNeg() || n.cmpn(3) < 0) return false;
    var one = new BN(1).toRed(curve.red);
    if (e.beta.cmp(one) === 0 || e.beta.redSqr().redMul(e.beta).cmp(one) !== 0) return false;
    // #E(F_p) = n: n is prime, lies in the Hasse interval and is wider than the interval
    var t = n.sub(p).isubn(1);
    if (t.sqr().cmp(p.muln(4)) > 0 || n.sqr().cmp(p.muln(16)) <= 0 || !probablyPrime(n)) return false;
    for (var i = 0; i < 2; i++) {
      var v = e.basis[i];
      if (!v || !BN.isBN(v.a) || !BN.isBN(v.b) || v.a.add(v.b.mul(e.lambda)).umod(n).cmpn(0) !== 0) return false;
    }
    var g0 = curve.point(g.x, g.y);                       // no tables: the plain ladder builds its own
    if (!curve.validate(g0)) return false;
    if (!orig.wnafMul.call(curve, g0, n).isInfinity()) return false;
    var lg = orig.wnafMul.call(curve, g0, e.lambda.umod(n));
    return !lg.isInfinity() && lg.x.cmp(g.x.redMul(e.beta)) === 0 && lg.y.cmp(g.y) === 0;
  }
  // the witness of that verdict: beta, lambda, both basis vectors, and n and G once more (the
  // verdict is about THIS generator and order)
  function snapEndo(curve) {
    var e = curve.endo, v = [];
    snapBN(v, e.beta); snapBN(v, e.lambda);
    for (var i = 0; i < 2; i++) { var b = Array.isArray(e.basis) ? e.basis[i] : null; snapBN(v, b && b.a); snapBN(v, b && b.b); }
    snapBN(v, curve.n);
    snapEntry(curve, v, curve.g);
    return Int32Array.from(v);
  }
  function sameEndo(curve, s) {
    var e = curve.endo, pos = sameBN(s, 0, e.beta, curve.red);
    if (pos >= 0) pos = sameBN(s, pos, e.lambda, null);
    if (pos >= 0 && (!Array.isArray(e.basis) || e.basis.length !== 2)) pos = -1;
    for (var i = 0; pos >= 0 && i < 2; i++) {
      var b = e.basis[i];
      pos = b ? sameBN(s, pos, b.a, null) : -1;
      if (pos >= 0) pos = sameBN(s, pos, b.b, null);
    }
    if (pos >= 0) pos = sameBN(s, pos, curve.n, null);
    if (pos >= 0) pos = sameEntry(curve, s, pos, curve.g);
    return pos === s.length;
  }
  function endoOK(curve) {
    try { return endoOK0(curve); } catch (e) { return false; }
  }
  function endoOK0(curve) {
    var e = curve.endo;
    if (!e) return true;
    var c = recall(curve, '_ellgpuEndo');
    if (c && c.red === curve.red && sameEndo(curve, c.snap)) return c.ok;
    var ok = false;
    refOnly++;
    try { ok = checkEndo(curve); } catch (x) { ok = false; } finally { refOnly--; }
    var snap = null;
    try { snap = snapEndo(curve); } catch (x) { snap = null; }
    if (snap) hide(curve, '_ellgpuEndo', { red: curve.red, snap: snap, ok: ok });
    return ok;
  }
  // the engine's domain for the ladders of this curve, or null: the reference's own code
  function ladderDomain(curve) {
    if (curve.type === 'mont') return null;
    var d = domain(curve) || customDomain(curve);
    if (d && curve.type === 'short' && curve.endo && !endoOK(curve)) return null;
    return d;
  }
  // ... and for the protocol calls, which also multiply the curve's own G (with ITS tables)
  function protocolDomain(curve) {
    var d = protocolDomainLazy(curve);
    return d && tablesOK(curve, curve.g) ? d : null;
  }
  // (the same without G's tables: the one-item calls look at those while the device works -- guarded)
  function protocolDomainLazy(curve) {
    var d = domain(curve);
    if (!d) return null;
    if (curve.type === 'short' && curve.endo && !endoOK(curve)) return null;
    return d;
  }
  // an EC / EDDSA instance multiplies ITS g and reduces by ITS n (ec/index.js:37-45, eddsa/index.js:
  // 17-20): they are the curve's unless somebody replaced them
  // (an EDDSA instance also carries its hash, the encoding length and the point class: eddsa/index.js:22-24)
  function ecOK(ec) {
    try { return ecOK0(ec); } catch (e) { return false; }
  }
  function ecOK0(ec) {
    var c = ec.curve;
    if (ec.g !== c.g) return false;
    if (ec.n === undefined) {                                  // EDDSA keeps no n of its own
      if (!HASH512) HASH512 = new elliptic.eddsa('ed25519').hash;
      return ec.hash === HASH512 && ec.encodingLength === 32 && ec.pointClass === c.point().constructor;
    }
    var n = c.n, h = ec.nh;
    if (ec.n !== n || !h || h.red || h.negative !== 0 || !n || n.negative !== 0) return false;
    // nh = n >> 1, word by word (26-bit words, dist/elliptic.js:3998): no allocation on this path
    var nw = n.words, hw = h.words, nl = n.length, hl = h.length;
    if (hl !== nl && hl !== nl - 1) return false;
    for (var i = 0; i < nl; i++) {
      var want = (nw[i] >>> 1) | (i + 1 < nl ? (nw[i + 1] & 1) << 25 : 0);
      if ((i < hl ? hw[i] : 0) !== want) return false;
    }
    return true;
  }
  // ---- the one-item calls: validate the tables WHILE the device computes ------------------------
  // call() is ONE engine call of a few items.  When any of `pts` carries precomputed tables, the
  // call is issued in its split form (ellgpu_ctx_defer: it returns once its work is enqueued), the
  // tables are compared with their witnesses (or checked from scratch the first time) while the
  // device works, and ellgpu_ctx_collect fetches the result -- which is DISCARDED (null: the caller
  // runs the reference's own method) when a table is not its point's multiples.
  // (ELLGPU_NO_DEFER=1, developer switch: validate first, then call -- what the overlap saves shows
  // as the difference, tools/bench_js_single_call.js)
  // RULE for call(): it returns the addon's result object as it is -- every result Buffer stays
  // referenced, and nothing is read out of them, until guarded() has collected.
  var canDefer = typeof addon.defer === 'function' && typeof addon.collect === 'function' && !eng.devices &&
    !process.env.ELLGPU_NO_DEFER;
  function guarded(curve, pts, call) {
    var any = false;
    for (var i = 0; i < pts.length; i++) if (pts[i] && pts[i].precomputed) any = true;
    if (!any) return call();
    function ok() {
      for (var j = 0; j < pts.length; j++) if (pts[j] && pts[j].precomputed && !tablesOK(curve, pts[j])) return false;
      return true;
    }
    if (!canDefer) return ok() ? call() : null;
    var r = null, good = false;
    addon.defer(eng.ctx);
    try { r = call(); good = ok(); } finally { addon.collect(eng.ctx); }
    return good ? r : null;
  }
  // The reference's GLV ladder fills a table-carrying operand's `precomputed.beta` the first time it
  // meets it (short.js:225 -> 282-310: lambda * P and ALL its table entries, from curve.endo.beta as
  // it is at that moment) -- state that its later calls read.  A call the engine answers in the GLV
  // ladder's place leaves that state behind as the reference's call would have: with the reference's
  // own _getBeta, at the same moment.  (Otherwise an edit of curve.endo.beta or of G's coordinates
  // between two calls meets a cache made at another time than the reference's: found by
  // tools/probe_mutation_walk.js.)  false: _getBeta threw -- what it half-did is taken back, and the
  // caller runs the reference's own method, which throws the same and leaves the same behind.
  // (... and false where the ladder's scratch arrays are no arrays any more: base.js:128-136,
  // short.js:218-221 write into curve._wnafT1..4 / _endoWnafT1..2 -- a TypeError of the reference's own)
  function scratchOK(curve) {
    return Array.isArray(curve._wnafT1) && Array.isArray(curve._wnafT2) && Array.isArray(curve._wnafT3) && Array.isArray(curve._wnafT4) &&
      (!curve.endo || (Array.isArray(curve._endoWnafT1) && Array.isArray(curve._endoWnafT2)));
  }
  function lazyBeta(curve, pts) {
    if (!scratchOK(curve)) return false;
    if (!curve.endo) return true;
    for (var i = 0; i < pts.length; i++) {
      var p = pts[i], pre = p && p.precomputed;
      if (!pre || typeof pre !== 'object' || pre.beta || typeof p._getBeta !== 'function') continue;
      var was = pre.beta;
      try { p._getBeta(); } catch (e) { try { pre.beta = was; } catch (x) { /* frozen */ } return false; }
    }
    return true;
  }
  // an operand as the engine takes it, or null where the reference must compute by itself
  function operandBuf(curve, p, B) {
    var b = affineBuf(curve, p, B);
    return b && tablesOK(curve, p) ? b : null;
  }
  function resultPoint(curve, d, r, jacobian) {
    var B = d.B;
    var pt;
    if (curve.type === 'short') {
      pt = r.inf[0] ? curve.point(null, null) :
        curve.point(new BN(r.xy.slice(0, B)), new BN(r.xy.slice(B, 2 * B)));
      return jacobian ? pt.toJ() : pt;
    }
    // edwards: identity is an ordinary point
    return curve.point(new BN(r.xy.slice(0, B)), new BN(r.xy.slice(B, 2 * B)));
  }

  var base = elliptic.curve.base.prototype;
  var short = elliptic.curve.short.prototype;
  var orig = {
    fixedNafMul: base._fixedNafMul,
    wnafMul: base._wnafMul,
    wnafMulAdd: base._wnafMulAdd,
    endoWnafMulAdd: short._endoWnafMulAdd,
  };

  // status 2 of the engine (ELLGPU_STATUS_OFF_CURVE): an operand is not on the curve.  The
  // reference computes with such points all the same; its own method gives its own answer.
  var OFF_CURVE = Engine.OFF_CURVE;
  // > 0 while the reference's own method runs on such an item: the ladders it calls internally
  // (_endoWnafMulAdd -> _wnafMulAdd) are the patched ones, and must not ask the engine again
  var refOnly = 0;
  function offCurve(curve, origFn, origArgs) {
    eng.stats.offCurve++; eng.stats.passthrough++;
    refOnly++;
    try { return origFn.apply(curve, origArgs); } finally { refOnly--; }
  }
  function mul1(curve, p, k, origFn, origArgs, glv) {
    var d = ladderDomain(curve);
    var kb = d && scalarBuf(k, d.B);
    var pb = kb && affineBuf(curve, p, d.B);
    if (pb && glv && !lazyBeta(curve, [p])) pb = null;
    var r = pb && guarded(curve, [p], function() {
      return isG(curve, d, p) ? eng.mulBatch(d.id, kb, null) : eng.mulBatch(d.id, kb, pb);
    });
    if (!r) { eng.stats.passthrough++; return origFn.apply(curve, origArgs); }
    if (r.inf[0] === OFF_CURVE) return offCurve(curve, origFn, origArgs);
    return resultPoint(curve, d, r, false);
  }
  function mulAdd(curve, p1, k1, p2, k2, jacobian, origFn, origArgs, glv) {
    var d = ladderDomain(curve);
    var b1 = d && scalarBuf(k1, d.B);
    var b2 = b1 && scalarBuf(k2, d.B);
    var q1 = b2 && affineBuf(curve, p1, d.B);
    var q2 = q1 && affineBuf(curve, p2, d.B);
    if (q2 && !(glv ? lazyBeta(curve, [p1, p2]) : scratchOK(curve))) q2 = null;
    var r = q2 && guarded(curve, [p1, p2], function() {
      return eng.mulAddBatch(d.id, b1, isG(curve, d, p1) ? null : q1, b2, q2);
    });
    if (!r) { eng.stats.passthrough++; return origFn.apply(curve, origArgs); }
    if (r.inf[0] === OFF_CURVE) return offCurve(curve, origFn, origArgs);
    return resultPoint(curve, d, r, jacobian);
  }

  // sum_i k_i * P_i for 3..8 points (the general form of _wnafMulAdd, base.js:128-253 -- even
  // counts only, as in the reference -- and of _endoWnafMulAdd, short.js:218-249; the reference's
  // own callers pass at most two points): the points
  // are paired up, every pair is one item of ONE k1*P1 + k2*P2 launch, and the partial sums are
  // added with the reference's Point#add.
  function mulAddMany(curve, points, coeffs, len, jacobian, origFn, origArgs, glv) {
    var d = ladderDomain(curve);
    var k1 = [], p1 = [], k2 = [], p2 = [];
    var ok = !!d && len >= 3 && len <= 8;
    if (ok && !glv) ok = scratchOK(curve);
    if (ok && glv) {
      for (var c = 0; ok && c < len; c++) ok = !!scalarBuf(coeffs[c], d.B) && !!affineBuf(curve, points[c], d.B);
      ok = ok && lazyBeta(curve, Array.prototype.slice.call(points, 0, len));
    }
    for (var i = 0; ok && i < len; i += 2) {
      var j = i + 1 < len ? i + 1 : i;               // odd tail: k * P + 0 * P
      var a = scalarBuf(coeffs[i], d.B), pa = operandBuf(curve, points[i], d.B);
      var b = j === i ? Buffer.alloc(d.B) : scalarBuf(coeffs[j], d.B);
      var pb = operandBuf(curve, points[j], d.B);
      if (!a || !pa || !b || !pb) { ok = false; break; }
      k1.push(a); p1.push(pa); k2.push(b); p2.push(pb);
    }
    if (!ok) { eng.stats.passthrough++; return origFn.apply(curve, origArgs); }
    var r = eng.mulAddBatch(d.id, Buffer.concat(k1), Buffer.concat(p1), Buffer.concat(k2),
      Buffer.concat(p2));
    for (var u = 0; u < k1.length; u++)
      if (r.inf[u] === OFF_CURVE) return offCurve(curve, origFn, origArgs);
    var acc = null;
    for (var t = 0; t < k1.length; t++) {
      var pt = resultPoint(curve, d, { xy: r.xy.slice(t * 2 * d.B, (t + 1) * 2 * d.B),
        inf: r.inf.slice(t, t + 1) }, false);
      acc = acc ? acc.add(pt) : pt;
    }
    return jacobian && acc.toJ ? acc.toJ() : acc;
  }

  base._fixedNafMul = function _fixedNafMul(p, k) {
    // (base.js:52-54: the reference asserts p.precomputed and reads its doubles)
    if (refOnly || !p || !p.precomputed || !p.precomputed.doubles) return orig.fixedNafMul.apply(this, arguments);
    return mul1(this, p, k, orig.fixedNafMul, arguments);
  };
  base._wnafMul = function _wnafMul(p, k) {
    if (refOnly) return orig.wnafMul.apply(this, arguments);
    return mul1(this, p, k, orig.wnafMul, arguments);
  };
  // The reference's pairing loop (base.js:165-183) takes two points whose NAF window is 1 -- points
  // without precomputed tables, `_getNAFPoints(defW = 1)` -- through `points[a].toJ()`, which
  // Edwards points do not have: `fresh.mulAdd(k1, fresh2, k2)` on an Edwards curve THROWS
  // "points[a].toJ is not a function" there (only a precomputed operand, e.g. EDDSA's G, makes
  // it work).  Such calls go to the original method, so the caller sees that same TypeError.
  function nafWindow(p, defW) {
    return p && p.precomputed && p.precomputed.naf ? p.precomputed.naf.wnd : defW;
  }
  function referenceThrowsOnPair(points, len, defW) {
    if (!points[0] || typeof points[0].toJ === 'function') return false;
    for (var i = len - 1; i >= 1; i -= 2)
      if (nafWindow(points[i - 1], defW) === 1 && nafWindow(points[i], defW) === 1) return true;
    return false;
  }
  base._wnafMulAdd = function _wnafMulAdd(defW, points, coeffs, len,
    jacobianResult) {
    if (refOnly) return orig.wnafMulAdd.apply(this, arguments);
    if (referenceThrowsOnPair(points, len, defW)) {
      eng.stats.passthrough++;
      return orig.wnafMulAdd.apply(this, arguments);
    }
    // (an odd len > 1 never worked in the reference: its pairing loop leaves naf[0] unset)
    if (len > 2 && len % 2 === 0)
      return mulAddMany(this, points, coeffs, len, !!jacobianResult, orig.wnafMulAdd, arguments);
    if (len !== 2) { eng.stats.passthrough++; return orig.wnafMulAdd.apply(this, arguments); }
    return mulAdd(this, points[0], coeffs[0], points[1], coeffs[1], !!jacobianResult,
      orig.wnafMulAdd, arguments);
  };
  short._endoWnafMulAdd = function _endoWnafMulAdd(points, coeffs,
    jacobianResult) {
    if (refOnly) return orig.endoWnafMulAdd.apply(this, arguments);
    if (points.length === 1) {
      var r = mul1(this, points[0], coeffs[0], orig.endoWnafMulAdd, arguments, true);
      return jacobianResult && r.toJ ? r.toJ() : r;
    }
    if (points.length === 2)
      return mulAdd(this, points[0], coeffs[0], points[1], coeffs[1],
        !!jacobianResult, orig.endoWnafMulAdd, arguments, true);
    return mulAddMany(this, points, coeffs, points.length, !!jacobianResult,
      orig.endoWnafMulAdd, arguments, true);
  };

  // point decompression: ShortCurve#pointFromX (short.js:187-204) and
  // EdwardsCurve#pointFromY (edwards.js:71-97).  An input with no point is handed to the
  // reference's own method, so the caller sees exactly the reference's exception ('invalid
  // point', or 'Assertion failed' out of bn.js's Tonelli-Shanks loop for 2^255 - 19)
  orig.pointFromX = short.pointFromX;
  short.pointFromX = function pointFromX(x, odd) {
    var d = domain(this);
    var xb = new BN(x, 16);
    var foreign = !!xb.red && xb.red !== this.red;      // the reference's first mixed operation throws
    if (xb.red) xb = xb.fromRed();
    if (!d || foreign || xb.isNeg() || xb.byteLength() > d.B) {
      eng.stats.passthrough++;
      return orig.pointFromX.apply(this, arguments);
    }
    var r = eng.decompressBatch(d.id, Buffer.from(xb.toArray('be', d.B)), Buffer.from([odd ? 1 : 0]));
    if (!r.ok[0]) return orig.pointFromX.apply(this, arguments);      // throws as the reference does
    return this.point(new BN(r.xy.slice(0, d.B)), new BN(r.xy.slice(d.B, 2 * d.B)));
  };
  var edw = elliptic.curve.edwards.prototype;
  orig.pointFromY = edw.pointFromY;
  edw.pointFromY = function pointFromY(y, odd) {
    var d = domain(this);
    var yb = new BN(y, 16);
    var foreign = !!yb.red && yb.red !== this.red;      // the reference's first mixed operation throws
    if (yb.red) yb = yb.fromRed();
    if (!d || foreign || yb.isNeg() || yb.byteLength() > d.B) {
      eng.stats.passthrough++;
      return orig.pointFromY.apply(this, arguments);
    }
    var r = eng.decompressBatch(d.id, Buffer.from(yb.toArray('be', d.B)), Buffer.from([odd ? 1 : 0]));
    if (!r.ok[0]) return orig.pointFromY.apply(this, arguments);      // throws as the reference does
    return this.point(new BN(r.xy.slice(0, d.B)), new BN(r.xy.slice(d.B, 2 * d.B)));
  };

  // EdwardsCurve#pointFromX (edwards.js:50-69): bit 1 of the parity byte selects it on the device
  orig.edPointFromX = edw.pointFromX;
  edw.pointFromX = function pointFromX(x, odd) {
    var d = domain(this);
    var xb = new BN(x, 16);
    var foreign = !!xb.red && xb.red !== this.red;      // the reference's first mixed operation throws
    if (xb.red) xb = xb.fromRed();
    if (!d || foreign || xb.isNeg() || xb.byteLength() > d.B) {
      eng.stats.passthrough++;
      return orig.edPointFromX.apply(this, arguments);
    }
    var r = eng.decompressBatch(d.id, Buffer.from(xb.toArray('be', d.B)), Buffer.from([odd ? 3 : 2]));
    if (!r.ok[0]) return orig.edPointFromX.apply(this, arguments);    // throws as the reference does
    return this.point(new BN(r.xy.slice(0, d.B)), new BN(r.xy.slice(d.B, 2 * d.B)));
  };

  // EC#recoverPubKey (ec/index.js:231-259): decompression of R, r^-1, both scalars and
  // s1*G + s2*R in one call.  Anything the engine does not take (toy curves, r outside
  // [1, n), digests longer than twice the order, non-byte messages) and every case where the
  // reference throws is run by the reference's own method, so results and exceptions are its own.
  var ecProto = elliptic.ec.prototype;
  // EC#sign (ec/index.js:110-186) with its default nonce source: the HmacDRBG draws, k*G, and
  // s = k^-1 (z + r d) in one call.  Byte-array digests only (the reference also takes hex
  // strings, numbers and BNs with their own length rules -- those, options.k / options.pers,
  // and anything the engine refuses go to the reference).
  // options.msgBitLength as the engine takes it (0 = eight bits per byte).  _truncateToN
  // (ec/index.js:81-108) uses ANY number it is given -- 0, a negative, a fraction, NaN: no shift, or
  // an assertion inside bn.js -- so only positive integers go to the engine; anything else that is
  // a number stays with the reference.
  function plainMsgBits(options) {
    if (!options || typeof options.msgBitLength !== 'number') return true;
    return Number.isInteger(options.msgBitLength) && options.msgBitLength > 0 && options.msgBitLength < (1 << 30);
  }
  var SigCtor = null;
  orig.sign = ecProto.sign;
  ecProto.sign = function sign(msg, key, enc, options) {
    if (typeof enc === 'object') { options = enc; enc = null; }
    if (!options) options = {};
    var d = protocolDomainLazy(this.curve);
    var res = null, self = this;
    try {
      if (!d || !ecOK(this) || options.k || options.pers !== undefined || typeof msg !== 'object' || BN.isBN(msg) ||
          !msg || typeof msg.length !== 'number' || msg.length === 0 || !plainMsgBits(options)) throw null;
      // new EC({ curve, hash }) may carry another DRBG hash than the preset's (ec/index.js:31)
      if (this.hash !== d.hash) throw null;                 // (the preset's hash as install() found it)
      for (var i = 0; i < msg.length; i++) if ((msg[i] & 255) !== msg[i]) throw null;
      var priv = this.keyFromPrivate(key, enc).getPrivate();
      var NB = this.n.byteLength();
      // a KeyPair is taken as it is (ec/key.js:31-32), so one made by ANOTHER EC instance may carry
      // a private key that was reduced by another order: the reference seeds its DRBG with those
      // bytes unreduced (ec/index.js:133), the engine reduces mod n first -- the reference's own
      if (priv.isNeg() || priv.cmp(this.n) >= 0) throw null;
      // the reference writes the truncated digest on n.byteLength() bytes (ec/index.js:136) and
      // throws 'byte array longer than desired length' when it does not fit -- p521 with a 67- or
      // 68-byte digest and an options.msgBitLength that shifts it by less than its excess
      if (this._truncateToN(msg, false, options.msgBitLength).byteLength() > NB) throw null;
      res = guarded(this.curve, [this.curve.g], function() {
        return eng.ecdsaSignDetBatch(d.id, { hashes: Buffer.from(msg), hashLen: msg.length,
          msgBits: typeof options.msgBitLength === 'number' ? options.msgBitLength : 0,
          priv: Buffer.from(priv.toArray('be', NB)), canonical: !!options.canonical });
      });
      if (!res || !res.ok[0]) throw null;
    } catch (e) {
      eng.stats.passthrough++;
      return orig.sign.apply(this, arguments);
    }
    // (the Signature class is not exported: it is taken from a signature the CALLER's own EC
    // instance makes with the original method -- constructing another EC here would give that
    // preset's shared G its tables as a side effect, which the unpatched library does not do)
    if (!SigCtor) SigCtor = orig.sign.call(this, [ 1 ], '01', 'hex').constructor;
    return new SigCtor({ r: new BN(res.r), s: new BN(res.s), recoveryParam: res.recid[0] });
  };
  // EC#verify (ec/index.js:188-229) as ONE engine call: s^-1, u1, u2, the double-scalar
  // multiplication and the comparison on the device (ellgpu_ecdsa_verify; a lone call runs as
  // three waves, csrc/work.h ecdsa_half) -- through the patched Point#mulAdd alone the inversion
  // and the two products mod n stay in JavaScript and the call costs MORE than the unpatched
  // library's (1.04 against 0.90 ms on the GPU box's host, profiles/r04_js_single_call.jsonl).
  // Decoding is the reference's own (keyFromPublic, Signature); byte-array digests on the preset
  // short curves only; whatever throws on the way, and a key that is not on the curve, goes to
  // the original method, which throws / answers by itself.
  orig.verify = ecProto.verify;
  ecProto.verify = function verify(msg, signature, key, enc, options) {
    var d = refOnly ? null : protocolDomainLazy(this.curve);
    if (!d || d.custom || this.curve.type !== 'short' || !ecOK(this) || !byteMessage(msg) || !plainMsgBits(options))
      return orig.verify.apply(this, arguments);
    var m, ok;
    try {
      var kp = this.keyFromPublic(key, enc);
      var pub = kp.getPublic();
      if (!pub || pub.isInfinity() || pub.curve !== this.curve) throw null;
      var item = { msg: msg, signature: signature, key: kp, options: options || undefined };
      m = marshalOne(this, d, item, true);
      if (m.ref) throw null;                   // a key the engine does not take (see marshalOne)
      // (ec/index.js:202-225: past its range checks the reference multiplies through the GLV ladder)
      if (m.pre && !lazyBeta(this.curve, [this.curve.g, pub])) throw null;
      var pk = packVerify([ m ], msg.length, msgBitsOf(item)).o;
      // G's tables, and the key's if it has any: looked at while the device works (the result's two
      // Buffers are filled when guarded() collects: read only after it has returned)
      var res = guarded(this.curve, [this.curve.g, pub], function() { return eng.ecdsaVerifyRaw(d.id, pk); });
      if (!res) throw null;                    // a table that is not its point's multiples
      ok = res.ok[0];
      if (res.status[0] === OFF_CURVE) ok = OFF_CURVE;
    } catch (e) {
      eng.stats.passthrough++;
      return orig.verify.apply(this, arguments);
    }
    if (ok === OFF_CURVE) return offCurve(this, orig.verify, arguments);
    return m.pre && ok === 1;
  };
  orig.recoverPubKey = ecProto.recoverPubKey;
  ecProto.recoverPubKey = function recoverPubKey(msg, signature, j, enc) {
    var d = protocolDomainLazy(this.curve);
    var e, r, s, NB;
    try {
      if (!d || !ecOK(this) || (3 & j) !== j) throw null;
      // the reference's Signature class is not exported; {r, s} objects (which include its own
      // instances) are decoded as it does (signature.js:20-21), DER input goes to the reference
      if (!signature || signature.r === undefined || signature.s === undefined) throw null;
      r = new BN(signature.r, 16);
      s = new BN(signature.s, 16);
      e = new BN(msg);
      if (r.red || s.red || e.red) throw null;
      NB = this.n.byteLength();
      if (e.isNeg() || r.isNeg() || s.isNeg() || r.byteLength() > NB || s.byteLength() > NB ||
          e.byteLength() > 2 * NB) throw null;
    } catch (x) {
      eng.stats.passthrough++;
      return orig.recoverPubKey.apply(this, arguments);
    }
    var hl = Math.max(e.byteLength(), 1);
    var res = guarded(this.curve, [this.curve.g], function() {
      return eng.ecdsaRecoverBatch(d.id, { hashes: Buffer.from(e.toArray('be', hl)), hashLen: hl,
        r: Buffer.from(r.toArray('be', NB)), s: Buffer.from(s.toArray('be', NB)), recid: Buffer.from([j]) });
    });
    if (!res) { eng.stats.passthrough++; return orig.recoverPubKey.apply(this, arguments); }
    // status 2 / 3: the reference throws, or inverts an unreduced r -- let it
    if (res.status[0] >= 2) return orig.recoverPubKey.apply(this, arguments);
    // (ec/index.js:252: g.mulAdd(s1, r, s2), reached once r has its point)
    if (!lazyBeta(this.curve, [this.curve.g])) return orig.recoverPubKey.apply(this, arguments);
    if (res.status[0] === 1) return this.curve.point(null, null);
    return this.curve.point(new BN(res.xy.slice(0, d.B)), new BN(res.xy.slice(d.B, 2 * d.B)));
  };

  // EDDSA#verify (eddsa/index.js:52-63) for ed25519: one launch does SHA-512, both point
  // decodings, S*G, h*A and the comparison.  Same results: false for S >= n; where the
  // reference throws (an undecodable R or A) its own method is run to throw the same Error.
  var eddsaProto = elliptic.eddsa.prototype;
  orig.eddsaVerify = eddsaProto.verify;
  // The reference computes with the OBJECTS it is given: a point passed as the key or as R is used
  // as it is (eddsa/key.js:20-23, eddsa/signature.js:33-38, eddsa/index.js:60-62) -- on the curve
  // or not, normalised or not -- a BN passed as S is S whatever `Sencoded` says, and `Rencoded` is
  // what gets hashed whatever R is.  The engine decodes R and A from their 32-byte encodings, so it
  // answers for the reference only when every object agrees with the encoding the engine is given:
  // same curve object, Z = 1, T = X Y, on the curve, and encodePoint(point) equal to the bytes.
  // Anything else is the reference's own.
  function sameBytes(a, b) {
    if (!a || !b || a.length !== b.length) return false;
    for (var i = 0; i < a.length; i++) if (a[i] !== b[i]) return false;
    return true;
  }
  function byteArray(a, len) {
    if (!a || typeof a.length !== 'number' || (len !== undefined && a.length !== len)) return false;
    for (var i = 0; i < a.length; i++) if ((a[i] & 255) !== a[i]) return false;
    return true;
  }
  function pointIsItsEncoding(eddsa, P, enc) {
    if (!P || P.curve !== eddsa.curve || !P.zOne || !P.x || !P.y || !P.t) return false;
    if (!P.x.red || !P.y.red || !P.t.red || P.t.cmp(P.x.redMul(P.y)) !== 0) return false;
    return eddsa.curve.validate(P) && tablesOK(eddsa.curve, P) && sameBytes(eddsa.encodePoint(P), enc);
  }
  eddsaProto.verify = function verify(message, sig, pub) {
    var d = protocolDomainLazy(this.curve);
    var utils = elliptic.utils;
    var m, sb, pb;
    try {
      if (!d || d.name !== 'ed25519' || !ecOK(this)) throw null;
      var mm = utils.parseBytes(message);
      // (hash.js takes array elements as they are; Buffer.from would reduce them mod 256)
      if (!byteArray(mm)) throw null;
      m = Buffer.from(mm);
      var sg = this.makeSignature(sig);
      if (sg.eddsa !== this && (!sg.eddsa || sg.eddsa.curve !== this.curve)) throw null;
      var S = sg.S();
      // eddsa/index.js:55-57, before the key is looked at
      if (!BN.isBN(S)) throw null;
      if (S.gte(this.curve.n) || S.isNeg()) return false;
      var key = this.keyFromPublic(pub);
      if (key.eddsa !== this && (!key.eddsa || key.eddsa.curve !== this.curve)) throw null;
      var renc = sg.Rencoded(), aenc = key.pubBytes();
      // (true Arrays only: EDDSA#decodePoint does bytes.slice(...).concat(...), eddsa/index.js:103,
      // which a Buffer or a Uint8Array does not have -- the reference throws a TypeError there)
      if (!Array.isArray(renc) || !Array.isArray(aenc) || !byteArray(renc, 32) || !byteArray(aenc, 32)) throw null;
      // point objects (given, or cached by an earlier call): the reference adds / multiplies THEM
      if (sg._R !== undefined && !pointIsItsEncoding(this, sg._R, renc)) throw null;
      if (key._pub !== undefined && !pointIsItsEncoding(this, key._pub, aenc)) throw null;
      sb = Buffer.concat([Buffer.from(renc), Buffer.from(S.toArray('le', 32))]);
      pb = Buffer.from(aenc);
    } catch (e) {
      eng.stats.passthrough++;
      return orig.eddsaVerify.apply(this, arguments);
    }
    var r = guarded(this.curve, [this.curve.g], function() { return eng.eddsaVerifyBatch([m], sb, pb); });
    if (!r) { eng.stats.passthrough++; return orig.eddsaVerify.apply(this, arguments); }
    if (r.err[0]) return orig.eddsaVerify.apply(this, arguments);      // throws as the reference does
    return r.ok[0] === 1;
  };

  // EDDSA#sign (eddsa/index.js:32-50) for ed25519 and 32-byte secrets: both hashes, a*G, r*G and
  // S = r + h*a in one call; other secrets (any length is legal for the reference) pass through.
  orig.eddsaSign = eddsaProto.sign;
  eddsaProto.sign = function sign(message, secret) {
    var d = protocolDomainLazy(this.curve);
    try {
      if (!d || d.name !== 'ed25519' || !ecOK(this)) throw null;
      var mm = elliptic.utils.parseBytes(message);
      var ss = this.keyFromSecret(secret).secret();
      // (hash.js takes array elements as they are; Buffer.from would reduce them mod 256)
      if (!byteArray(mm) || !byteArray(ss, 32)) throw null;
      var m = Buffer.from(mm);
      var sec = Buffer.from(ss);
    } catch (e) {
      eng.stats.passthrough++;
      return orig.eddsaSign.apply(this, arguments);
    }
    var r = guarded(this.curve, [this.curve.g], function() { return eng.eddsaSignBatch([m], sec); });
    if (!r) { eng.stats.passthrough++; return orig.eddsaSign.apply(this, arguments); }
    return this.makeSignature(Array.prototype.slice.call(r.sig, 0, 64));
  };

  // Montgomery x-only ladder (Point class is not exported: reach it as
  // eddsa/index.js:22 does, through an instance)
  var montProto = elliptic.curves.curve25519.curve.g.constructor.prototype;
  orig.montMul = montProto.mul;
  montProto.mul = function mul(k) {
    var d = domain(this.curve);
    var kb = d && scalarBuf(k, 32);
    var mine = false;
    try { mine = !!kb && !this.isInfinity() && this.x.red === this.curve.red && this.z.red === this.curve.red &&
      this.x.negative === 0 && this.z.negative === 0; } catch (e) { mine = false; }
    if (!mine) {
      eng.stats.passthrough++;
      return orig.montMul.apply(this, arguments);
    }
    var x = this.curve.point(this.x, this.z).getX();
    var r = eng.x25519Batch(kb, Buffer.from(x.toArray('be', 32)));
    if (r.inf[0]) return this.curve.point(null, null);
    return this.curve.point(new BN(r.x), new BN(1));
  };

  // KeyPair#derive on curve25519 (ec/key.js:102-107): pub.validate() -- a square root in JavaScript,
  // mont.js:23-32, 290 us -- and pub.mul(priv).getX() as ONE engine call (validity by Euler's criterion
  // on a wave of its own beside the ladder's).  Everything else about the call is the reference's:
  // short curves (their validate is two products), points of another curve object, infinity, and --
  // through the original method -- the exception it throws where x has no point on the curve.
  var kpProto = new elliptic.ec('curve25519').keyFromPrivate('01', 'hex').constructor.prototype;
  orig.derive = kpProto.derive;
  kpProto.derive = function derive(pub) {
    var ec = this.ec, curve = ec && ec.curve;
    var mine = false;
    try { mine = !refOnly && curve && curve.type === 'mont' && pub && pub.curve === curve && this.priv &&
        typeof pub.isInfinity === 'function' && !pub.isInfinity() && pub.x && pub.z &&
        pub.x.red === curve.red && pub.z.red === curve.red && pub.x.negative === 0 && pub.z.negative === 0; } catch (e) { mine = false; }
    if (mine) {
      var d = domain(curve);
      var kb = d && scalarBuf(this.priv, 32);
      if (kb) {
        // (the reference's validate() normalizes the caller's point in place, mont.js:24, 160-165: so does this)
        var x = pub.normalize().x.fromRed();
        var r = eng.x25519DeriveBatch(kb, Buffer.from(x.toArray('be', 32)));
        if (r.status[0] === 0) return new BN(r.x);
      }
    }
    eng.stats.passthrough++;
    return orig.derive.apply(this, arguments);
  };

  // what install() takes a curve object for (tests; nothing is stored on the object itself):
  // { preset: name | null, custom: curve id | null }
  eng.recognised = function recognised(curve) {
    var d = domain(curve), c = d ? null : customDomain(curve);
    return { preset: d ? d.name : null, custom: c ? c.id : null };
  };

  eng.uninstall = function uninstall() {
    kpProto.derive = orig.derive;
    base._fixedNafMul = orig.fixedNafMul;
    base._wnafMul = orig.wnafMul;
    base._wnafMulAdd = orig.wnafMulAdd;
    short._endoWnafMulAdd = orig.endoWnafMulAdd;
    montProto.mul = orig.montMul;
    short.pointFromX = orig.pointFromX;
    eddsaProto.verify = orig.eddsaVerify;
    eddsaProto.sign = orig.eddsaSign;
    ecProto.recoverPubKey = orig.recoverPubKey;
    ecProto.verify = orig.verify;
    ecProto.sign = orig.sign;
    edw.pointFromY = orig.pointFromY;
    edw.pointFromX = orig.edPointFromX;
  };

  // EC#verify over many signatures with the reference's own decoding
  // (keyFromPublic, Signature, _truncateToN's length rule) and ONE launch.
  // items: [{ msg: Buffer|Array, signature, key, enc? }] -> [bool]
  // a preset whose G carries tables that are not G's multiples, or whose endomorphism constants are
  // not the curve's (protocolDomain): every item through EC#verify, whose ladders decide by themselves
  function untrusted(ec, d) { return d && !(lazyBeta(ec.curve, [ec.curve.g]) && protocolDomain(ec.curve) && ecOK(ec)); }
  function verifyEach(ec, items) {
    return items.map(function(it) { return ec.verify(it.msg, it.signature, it.key, it.enc, it.options); });
  }
  eng.verifyMany = function verifyMany(ec, items) {
    // (a curve object that is no preset -- or is one no longer --: EC#verify of every item, which is what
    // this call stands for in every case)
    var d = ec && ec.curve ? domain(ec.curve) : null;
    if (!d || ec.curve.type !== 'short' || untrusted(ec, d)) return verifyEach(ec, items);
    var m = marshalVerify(ec, d, items);
    m.o.status = Buffer.alloc(items.length);
    var ok = eng.ecdsaVerifyBatch(d.id, m.o);
    return items.map(function(it, i) { return verdict(ec, it, m.pre[i], m.ref[i] || m.o.status[i] === OFF_CURVE ? OFF_CURVE : ok[i]); });
  };
  // a key that is not on the curve (status 2): the reference computes with it -- and can answer
  // true -- so that item goes through EC#verify itself, with the reference's own ladders
  function verdict(ec, it, pre, ok) {
    if (pre && ok === OFF_CURVE) {
      eng.stats.offCurve++;
      refOnly++;
      try { return ec.verify(it.msg, it.signature, it.key, it.enc, it.options); } finally { refOnly--; }
    }
    return pre && ok === 1;
  }
  // Promise form of verifyMany: the batch runs on a libuv worker thread (ecdsaVerifyBatchAsync)
  // ec/signature.js is not exported by the library: its constructor is taken, once per
  // install(), from the first signature an EC instance makes
  var SignatureClass = null;
  function signatureClass(ec) {
    if (!SignatureClass) SignatureClass = ec.sign('00', ec.keyFromPrivate('01', 'hex')).constructor;
    return SignatureClass;
  }
  // one (msg, signature, key) as the engine's fixed-width fields; throws what EC#verify throws
  function marshalOne(ec, d, it, lazyTables) {
    var NB = ec.n.byteLength();
    var Signature = signatureClass(ec);
    var key = ec.keyFromPublic(it.key, it.enc);
    var sig = new Signature(it.signature, 'hex');
    // (r or s in a reduction context: eqXToP's r.toRed throws 'toRed works only with numbers')
    var redSig = !!(sig.r.red || sig.s.red);              // -- that item is the reference's
    var bad = sig.r.isNeg() || sig.s.isNeg() || sig.r.byteLength() > NB || sig.s.byteLength() > NB;
    // (a key whose own precomputed tables are not its multiples: the reference's ladder reads
    // them -- that item is the reference's, like a key that is not on the curve)
    // (lazyTables: the caller looks at the key's tables itself, while the device works -- guarded)
    var q = redSig ? null : (lazyTables ? affineBuf : operandBuf)(ec.curve, key.getPublic(), d.B);
    return { pre: !bad, h: Buffer.from(it.msg),
      r: Buffer.from((bad ? new BN(0) : sig.r).toArray('be', NB)),
      s: Buffer.from((bad ? new BN(0) : sig.s).toArray('be', NB)),
      q: q || Buffer.alloc(2 * d.B), ref: !q };   // ref: EC#verify itself, with the reference's own ladders (verdict)
  }
  function msgBitsOf(it) {
    return it.options && typeof it.options.msgBitLength === 'number' ? it.options.msgBitLength : 0;
  }
  function packVerify(ms, hl, msgBits) {
    return { pre: ms.map(function(m) { return m.pre; }), ref: ms.map(function(m) { return m.ref; }),
      o: { hashes: Buffer.concat(ms.map(function(m) { return m.h; })), hashLen: hl, msgBits: msgBits | 0,
        r: Buffer.concat(ms.map(function(m) { return m.r; })), s: Buffer.concat(ms.map(function(m) { return m.s; })),
        pub: Buffer.concat(ms.map(function(m) { return m.q; })) } };
  }
  function marshalVerify(ec, d, items) {
    var hl = items.length ? items[0].msg.length : 1;
    var mb = items.length ? msgBitsOf(items[0]) : 0;
    return packVerify(items.map(function(it) {
      if (it.msg.length !== hl || msgBitsOf(it) !== mb)
        throw new Error('verifyMany: digests must share one length (and one options.msgBitLength)');
      return marshalOne(ec, d, it);
    }), hl, mb);
  }
  eng.verifyManyAsync = function verifyManyAsync(ec, items) {
    var d = ec && ec.curve ? domain(ec.curve) : null;
    if (!d || ec.curve.type !== 'short' || untrusted(ec, d)) return new Promise(function(resolve) { resolve(verifyEach(ec, items)); });
    var m;
    try { m = marshalVerify(ec, d, items); } catch (e) { return Promise.reject(e); }
    if (!items.length) return Promise.resolve([]);
    m.o.status = Buffer.alloc(items.length);
    return eng.ecdsaVerifyBatchAsync(d.id, m.o).then(function(ok) {
      return items.map(function(it, i) { return verdict(ec, it, m.pre[i], m.ref[i] || m.o.status[i] === OFF_CURVE ? OFF_CURVE : ok[i]); });
    });
  };
  // One verification as a Promise -- and the answer to "one ec.verify is one launch of one lane"
  // (ec/index.js:188-229 costs the device ~1 ms per call, what the reference's own JavaScript
  // costs): every verifyAsync issued before the event loop turns joins ONE batch (a microtask
  // flushes the queue), so N concurrent callers share a launch instead of queueing N of them.
  // Digests of different lengths (or options.msgBitLength), or different EC instances, form
  // separate batches; a call whose key or signature the reference would throw on rejects with
  // that error, alone; a key that is not on the curve gets the reference's own verdict.
  var pendingVerify = [];
  function flushVerify() {
    var q = pendingVerify;
    pendingVerify = [];
    var groups = [];
    q.forEach(function(p) {
      var g = null;
      var mb = msgBitsOf(p.item);
      for (var i = 0; i < groups.length && !g; i++)
        if (groups[i].ec === p.ec && groups[i].hl === p.item.msg.length && groups[i].mb === mb) g = groups[i];
      if (!g) { g = { ec: p.ec, hl: p.item.msg.length, mb: mb, ps: [] }; groups.push(g); }
      g.ps.push(p);
    });
    groups.forEach(function(g) {
      var good = [], ms = [];
      var d = domain(g.ec.curve);
      function each(ps) {                  // every call by itself, through the (patched) synchronous path
        ps.forEach(function(p) {
          try { p.resolve(g.ec.verify(p.item.msg, p.item.signature, p.item.key, p.item.enc, p.item.options)); } catch (e) { p.reject(e); }
        });
      }
      if (!d) return each(g.ps);
      g.ps.forEach(function(p) {           // a throwing item rejects alone
        try { ms.push(marshalOne(g.ec, d, p.item)); good.push(p); }
        catch (e) { p.reject(e); }
      });
      if (!good.length) return;
      eng.stats.coalescedBatches = (eng.stats.coalescedBatches || 0) + 1;
      eng.stats.coalescedItems = (eng.stats.coalescedItems || 0) + good.length;
      var m = packVerify(ms, g.hl, g.mb);
      m.o.status = Buffer.alloc(good.length);
      var job = eng.ecdsaVerifyBatchAsync(d.id, m.o);
      // G's tables: looked at once per batch (the calls of one tick see one state of the library),
      // HERE -- in the tick of the calls, while the worker thread already runs the batch; a table
      // that is not G's multiples leaves every call to the synchronous path, the batch's verdicts unused
      var trusted = lazyBeta(g.ec.curve, [g.ec.curve.g]) && tablesOK(g.ec.curve, g.ec.curve.g);
      if (!trusted) { job.catch(function() {}); return each(good); }
      job.then(function(ok) {
        good.forEach(function(p, i) {
          try { p.resolve(verdict(g.ec, p.item, m.pre[i], m.ref[i] || m.o.status[i] === OFF_CURVE ? OFF_CURVE : ok[i])); } catch (e) { p.reject(e); }
        });
      }, function(e) { good.forEach(function(p) { p.reject(e); }); });
    });
  }
  // msg as the batch takes it: a non-empty array-like of BYTES (an Array with other elements goes
  // through the reference's `new BN(msg, 16)`, which does not truncate them mod 256; an empty
  // message is z = 0 there, not a zero-length digest)
  function byteMessage(msg) {
    if (!(Buffer.isBuffer(msg) || Array.isArray(msg) || msg instanceof Uint8Array) || msg.length === 0) return false;
    if (Array.isArray(msg))
      for (var i = 0; i < msg.length; i++) if ((msg[i] & 255) !== msg[i]) return false;
    return true;
  }
  eng.verifyAsync = function verifyAsync(ec, msg, signature, key, enc, options) {
    if (typeof enc === 'object' && enc !== null && options === undefined) { options = enc; enc = undefined; }
    var d = ecOK(ec) ? protocolDomainLazy(ec.curve) : null;        // (G's tables: flushVerify, once per batch)
    if (!d || ec.curve.type !== 'short' || !byteMessage(msg) || !plainMsgBits(options)) {
      // outside the engine's batch domain: the (patched) synchronous path, as a Promise
      return new Promise(function(resolve) { resolve(ec.verify(msg, signature, key, enc, options)); });
    }
    return new Promise(function(resolve, reject) {
      pendingVerify.push({ ec: ec, item: { msg: Buffer.from(msg), signature: signature, key: key, enc: enc,
        options: options }, resolve: resolve, reject: reject });
      if (pendingVerify.length === 1) Promise.resolve().then(flushVerify);
    });
  };
  return eng;
}

module.exports = { install: install, Engine: Engine, CURVES: CURVES };
