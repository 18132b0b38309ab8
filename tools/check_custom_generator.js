'use strict';
// install() must not serve curves that only SHARE a preset's field from the preset's tables:
// same p, a, b as secp256k1 / p256 but another generator (or none at all).  Such a curve is a
// user-defined curve: with options.customCurves === false it stays on the reference's own code
// (no engine call at all), by default it runs on the device's generic path (run-time prime, no
// fixed-base table, no GLV).  Either way the patched library's results are compared with an
// unpatched copy of the reference.
//   ELLGPU_LIB=<hostsim or real library> [ELLGPU_CUSTOM=0] node tools/check_custom_generator.js
var loader = require('./ref_loader');
var plain = loader.load().elliptic;          // unpatched
var patched = loader.load().elliptic;        // a second, independent copy -> patched
var CUSTOM = process.env.ELLGPU_CUSTOM !== '0';
var eng = require('../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB,
  customCurves: CUSTOM });
var BN = plain.curves.secp256k1.curve.p.constructor;
var checked = 0;

function same(a, b, what) {
  if (a.isInfinity() !== b.isInfinity()) throw new Error(what + ': infinity differs');
  if (!a.isInfinity() && (a.getX().cmp(b.getX()) !== 0 || a.getY().cmp(b.getY()) !== 0))
    throw new Error(what + ': ' + a.getX().toString(16) + ' != ' + b.getX().toString(16));
  checked++;
}

['secp256k1', 'p256'].forEach(function(name) {
  var ref = plain.curves[name].curve;
  var g2 = ref.g.mul(new BN(2));
  function build(lib, withG) {
    var conf = { p: ref.p.toString(16), a: ref.a.fromRed().toString(16), b: ref.b.fromRed().toString(16),
      n: ref.n.toString(16) };
    if (withG) conf.g = [ g2.getX().toString(16), g2.getY().toString(16) ];
    return new lib.curve.short(conf);
  }
  var before = eng.stats.gpuCalls;
  var cp = build(plain, true), cq = build(patched, true);
  [ '1', '2', 'deadbeef', ref.n.subn(1).toString(16), ref.n.toString(16),
    'ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad2' ].forEach(function(k) {
    same(cq.g.mul(new BN(k, 16)), cp.g.mul(new BN(k, 16)), name + ' custom g * ' + k);
    same(cq.g.mulAdd(new BN(k, 16), cq.g.dbl(), new BN(7)), cp.g.mulAdd(new BN(k, 16), cp.g.dbl(), new BN(7)),
      name + ' custom mulAdd ' + k);
  });
  // ECDSA over the custom generator: sign with the patched library, verify with both
  var ecq = new patched.ec({ curve: { curve: cq, g: cq.g, n: cq.n, hash: patched.curves[name].hash } });
  var ecp = new plain.ec({ curve: { curve: cp, g: cp.g, n: cp.n, hash: plain.curves[name].hash } });
  var key = ecq.keyFromPrivate('1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef', 'hex');
  var msg = new BN('7777777777777777777777777777777777777777777777777777777777777777', 16);
  var sig = ecq.sign(msg, key);
  var pub = key.getPublic();
  if (!ecq.verify(msg, sig, pub)) throw new Error(name + ': patched verify of its own signature');
  var pubp = ecp.keyFromPublic({ x: pub.getX().toString(16), y: pub.getY().toString(16) });
  if (!ecp.verify(msg, { r: sig.r.toString(16), s: sig.s.toString(16) }, pubp))
    throw new Error(name + ': the reference rejects the patched library\'s signature');
  checked += 2;
  // a curve without a generator: arbitrary points still multiply (no TypeError on curve.g.x)
  var np = build(plain, false), nq = build(patched, false);
  var x = ref.g.getX().toString(16), y = ref.g.getY().toString(16);
  same(nq.point(x, y).mul(new BN('c0ffee', 16)), np.point(x, y).mul(new BN('c0ffee', 16)), name + ' no generator');
  if (!CUSTOM && eng.stats.gpuCalls !== before)
    throw new Error(name + ': a custom-generator curve reached the engine');
  if (CUSTOM && eng.stats.gpuCalls === before)
    throw new Error(name + ': the user-defined curve did not reach the engine');
  // (what install() takes the curve object for -- kept in its own WeakMaps, nothing is written on the object)
  var rec = eng.recognised(cq);
  if (Object.keys(cq).some(function(k) { return /ellgpu/i.test(k); }) || Object.getOwnPropertyNames(cq).some(function(k) { return /ellgpu/i.test(k); }))
    throw new Error(name + ': install() wrote on the caller\'s curve object');
  if (CUSTOM && (rec.preset !== null || rec.custom === null || rec.custom < 16))
    throw new Error(name + ': a custom-generator curve was taken for the preset');
});
// the general forms of _wnafMulAdd / _endoWnafMulAdd: 3 and 4 points (paired up on the device)
['secp256k1', 'p256', 'p521'].forEach(function(name) {
  var cp = plain.curves[name].curve, cq = patched.curves[name].curve;
  var ks = ['3', 'deadbeefcafebabe0123456789abcdef', 'ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad',
    '1'].map(function(k) { return new BN(k, 16); });
  function pts(c) {
    return [3, 5, 7, 11].map(function(m) { return c.g.mul(new BN(m)); });
  }
  var b0 = eng.stats.gpuCalls;
  [3, 4].forEach(function(len) {
    [false, true].forEach(function(jac) {
      var a, b;
      if (len % 2 === 0) {              // (an odd count never worked in the reference itself)
        a = cq._wnafMulAdd(1, pts(cq).slice(0, len), ks.slice(0, len), len, jac);
        b = cp._wnafMulAdd(1, pts(cp).slice(0, len), ks.slice(0, len), len, jac);
        same(a.toP ? a.toP() : a, b.toP ? b.toP() : b, name + ' _wnafMulAdd ' + len);
      }
      if (name === 'secp256k1') {
        a = cq._endoWnafMulAdd(pts(cq).slice(0, len), ks.slice(0, len), jac);
        b = cp._endoWnafMulAdd(pts(cp).slice(0, len), ks.slice(0, len), jac);
        same(a.toP ? a.toP() : a, b.toP ? b.toP() : b, name + ' _endoWnafMulAdd ' + len);
      }
    });
  });
  // P + (-P) + Q: a partial sum at infinity
  var q = pts(cq), r = pts(cp);
  same(cq._wnafMulAdd(1, [q[0], q[0].neg(), q[1], q[2]], [ks[1], ks[1], ks[0], ks[2]], 4, false),
    cp._wnafMulAdd(1, [r[0], r[0].neg(), r[1], r[2]], [ks[1], ks[1], ks[0], ks[2]], 4, false), name + ' cancelling pair');
  if (eng.stats.gpuCalls === b0) throw new Error(name + ': the 3/4-point forms did not reach the engine');
});
// Edwards curves that are not ed25519 (projective coordinates for a != -1): Point#mul on the device
(function() {
  function build(lib) {
    var p = new BN(1).ushln(251).subn(9);                       // Curve1174
    return new lib.curve.edwards({ p: p.toString(16), a: '1', c: '1', d: p.subn(1174).toString(16) });
  }
  var cp = build(plain), cq = build(patched);
  var gp = null, gq = null;
  for (var y = 2; !gp; y++) {
    try { gp = cp.pointFromY(new BN(y), false); gq = cq.pointFromY(new BN(y), false); } catch (e) { gp = null; }
  }
  var b0 = eng.stats.gpuCalls;
  [ '1', '2', 'deadbeef', 'ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad2ab54a98ceb1f0ad',
    'ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff' ].forEach(function(k) {
    var a = gq.mul(new BN(k, 16)), b = gp.mul(new BN(k, 16));
    if (a.getX().cmp(b.getX()) !== 0 || a.getY().cmp(b.getY()) !== 0) throw new Error('curve1174 mul ' + k);
    if (!cq.validate(a)) throw new Error('curve1174: result not on the curve');
    checked++;
  });
  if (CUSTOM && eng.stats.gpuCalls === b0) throw new Error('curve1174 did not reach the engine');
  if (!CUSTOM && eng.stats.gpuCalls !== b0) throw new Error('curve1174 reached the engine');
})();
// the presets themselves still go to the engine
var b0 = eng.stats.gpuCalls;
same(patched.curves.secp256k1.curve.g.mul(new BN(5)), plain.curves.secp256k1.curve.g.mul(new BN(5)), 'preset');
if (eng.stats.gpuCalls === b0) throw new Error('the preset did not reach the engine');
console.log(JSON.stringify({ ok: true, custom: CUSTOM, checked: checked, engine: eng.stats }));
process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
