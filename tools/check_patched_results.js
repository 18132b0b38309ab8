'use strict';
// Differential check of the patched public API against the golden files (which were produced
// by the unpatched reference): EC#recoverPubKey, EDDSA#sign, EDDSA#verify and pointFromX /
// pointFromY through install(), including the MESSAGE of every exception the reference throws.
// Build container only (needs /root/reference).
//
//   ELLGPU_LIB=tests/hostsim/_build/libellgpu_hostsim.so node tools/check_patched_results.js
var fs = require('fs');
var path = require('path');
var ref = require('./ref_loader').load();
var elliptic = ref.elliptic;
var eng = require('../elliptic_amd/js').install(elliptic, { libPath: process.env.ELLGPU_LIB });
var GOLD = path.join(__dirname, '..', 'tests', 'golden');
function load(name) { return JSON.parse(fs.readFileSync(path.join(GOLD, name))); }
var checked = 0, thrown = 0;
function expectThrow(fn, msg, what) {
  try { fn(); } catch (e) {
    if (e.message !== msg) throw new Error(what + ': threw "' + e.message + '", the reference throws "' + msg + '"');
    thrown++;
    return;
  }
  throw new Error(what + ': did not throw, the reference throws "' + msg + '"');
}

['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521'].forEach(function(name) {
  var ec = new elliptic.ec(name);
  var L = ec.curve.p.byteLength();
  load('recover_' + name + '.json').forEach(function(c) {
    var z = Buffer.from(c.z, 'hex').toJSON().data;
    var run = function() { return ec.recoverPubKey(z, { r: c.r, s: c.s }, c.j); };
    if (c.throws) return expectThrow(run, c.throws, 'recoverPubKey ' + name + ' (' + c.note + ')');
    var q = run();
    if (c.q.inf ? !q.isInfinity() : (q.isInfinity() || q.getX().toString(16, 2 * L) !== c.q.x ||
        q.getY().toString(16, 2 * L) !== c.q.y))
      throw new Error('recoverPubKey mismatch: ' + name + ' ' + c.note);
    checked++;
  });
});

['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521'].forEach(function(name) {
  var ec = new elliptic.ec(name);
  var NB = ec.n.byteLength();
  load('signdet_' + name + '.json').forEach(function(c) {
    var sig = ec.sign(Buffer.from(c.z, 'hex').toJSON().data, c.d, 'hex', { canonical: c.canonical });
    if (sig.r.toString(16, 2 * NB) !== c.r || sig.s.toString(16, 2 * NB) !== c.s || sig.recoveryParam !== c.recid)
      throw new Error('sign mismatch: ' + name + ' ' + c.note);
    // the result must behave like the reference's Signature (DER round trip, verify)
    if (!ec.verify(Buffer.from(c.z, 'hex').toJSON().data, sig.toDER('hex'), ec.keyFromPrivate(c.d, 'hex')))
      throw new Error('signature does not verify: ' + name + ' ' + c.note);
    checked++;
  });
});

var ed = new elliptic.eddsa('ed25519');
load('eddsa_sign_ed25519.json').forEach(function(c) {
  var msg = c.msg.length ? Buffer.from(c.msg, 'hex').toJSON().data : [];
  var key = ed.keyFromSecret(c.secret);
  if (key.sign(msg).toHex().toLowerCase() !== c.sig) throw new Error('eddsa sign mismatch: ' + c.note);
  checked++;
});
load('eddsa_verify_ed25519.json').forEach(function(c) {
  var msg = c.msg.length ? Buffer.from(c.msg, 'hex').toJSON().data : [];
  var run = function() { return ed.verify(msg, c.sig, c.pub); };
  if (c.throws) return expectThrow(run, c.throws, 'eddsa verify (' + c.note + ')');
  if (run() !== c.ok) throw new Error('eddsa verify mismatch: ' + c.note);
  checked++;
});
['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521', 'ed25519'].forEach(function(name) {
  var curve = elliptic.curves[name].curve;
  var L = curve.p.byteLength();
  load('decompress_' + name + '.json').forEach(function(c) {
    var run = function() {
      return name === 'ed25519' ? curve.pointFromY(c.v, c.odd) : curve.pointFromX(c.v, c.odd);
    };
    if (c.r.invalid !== undefined) return expectThrow(run, c.r.invalid, 'decompress ' + name);
    var q = run();
    if (q.getX().toString(16, 2 * L) !== c.r.x || q.getY().toString(16, 2 * L) !== c.r.y)
      throw new Error('decompress mismatch: ' + name);
    checked++;
  });
});
// EdwardsCurve#pointFromX (edwards.js:50-69)
(function() {
  var curve = elliptic.curves.ed25519.curve;
  load('fromx_ed25519.json').forEach(function(c) {
    var run = function() { return curve.pointFromX(c.v, c.odd); };
    if (c.r.invalid !== undefined) return expectThrow(run, c.r.invalid, 'ed25519 pointFromX');
    var q = run();
    if (q.getX().toString(16, 64) !== c.r.x || q.getY().toString(16, 64) !== c.r.y)
      throw new Error('ed25519 pointFromX mismatch');
    checked++;
  });
})();
console.log(JSON.stringify({ ok: true, checked: checked, thrown: thrown, engine: eng.stats }));
