'use strict';
// Differential check of the patched public API against the golden files (which were produced
// by the unpatched reference): EC#recoverPubKey, EDDSA#sign, EDDSA#verify and pointFromX /
// pointFromY through install(), including the MESSAGE of every exception the reference throws,
// and the reference's answers on points that are not on the curve (offcurve_*.json).
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
// Edwards mulAdd / jmulAdd without a precomputed operand: the reference's pairing loop calls toJ(),
// which Edwards points lack (base.js:174-183) -- it throws, and so must the patched library
(function() {
  var c = elliptic.curves.ed25519.curve;
  var hadTables = c.g.precomputed;
  c.g.precomputed = null;                               // (an EDDSA instance may have tabled G already)
  var P = c.g.mul(new ref.BN(5)), Q = c.g.mul(new ref.BN(7));
  var P2 = c.point(P.getX(), P.getY()), Q2 = c.point(Q.getX(), Q.getY());
  expectThrow(function() { return P2.mulAdd(new ref.BN(3), Q2, new ref.BN(4)); }, 'points[a].toJ is not a function', 'edwards fresh.mulAdd(fresh)');
  expectThrow(function() { return c.g.mulAdd(new ref.BN(3), Q2, new ref.BN(4)); }, 'points[a].toJ is not a function', 'edwards untabled g.mulAdd');
  expectThrow(function() { return P2.jmulAdd(new ref.BN(3), Q2, new ref.BN(4)); }, 'points[a].toJ is not a function', 'edwards fresh.jmulAdd(fresh)');
  c.g.precomputed = hadTables;
  c.g.precompute(c.n.bitLength() + 1);
  var R = c.g.mulAdd(new ref.BN(3), Q2, new ref.BN(4));   // with a tabled G it works, and on the device
  var W = c.g.mul(new ref.BN(3)).add(Q2.mul(new ref.BN(4)));
  if (R.getX().cmp(W.getX()) !== 0 || R.getY().cmp(W.getY()) !== 0) throw new Error('edwards g.mulAdd mismatch');
  checked++;
})();

// Points that are not on the curve (offcurve_<curve>.json): the reference computes with them, the
// engine reports them with status 2, and install() hands those items to the reference's own
// method -- so the patched library must return the reference's answer: Point#mul, mulAdd /
// jmulAdd, EC#verify (incl. the tuples the reference answers TRUE on), verifyMany and the
// coalescing verifyAsync.
var pendingAsync = [];
['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521', 'ed25519'].forEach(function(name) {
  var pc = elliptic.curves[name];
  var curve = pc.curve;
  var L = curve.p.byteLength();
  var before = eng.stats.offCurve;
  var nOff = 0;
  function same(pt, r) {
    if (r.inf && r.x === undefined) return pt.isInfinity();
    return !pt.isInfinity() || name === 'ed25519' ? (pt.getX().toString(16, 2 * L) === r.x &&
      pt.getY().toString(16, 2 * L) === r.y) : false;
  }
  var ec = curve.type === 'short' ? new elliptic.ec(pc) : null;
  if (!ec) curve.g.precompute(curve.n.bitLength() + 1);           // as EDDSA's constructor does
  var items = [], wants = [];
  load('offcurve_' + name + '.json').forEach(function(c) {
    if (c.op === 'var') {
      if (!same(curve.point(c.px, c.py).mul(new ref.BN(c.k, 16)), c.r)) throw new Error('off-curve mul mismatch: ' + name);
      if (!c.on) nOff++;
    } else if (c.op === 'muladd') {
      var A = c.g1 ? curve.g : curve.point(c.p1x, c.p1y);
      var B = curve.point(c.p2x, c.p2y);
      if (!same(A.mulAdd(new ref.BN(c.k1, 16), B, new ref.BN(c.k2, 16)), c.r)) throw new Error('off-curve mulAdd mismatch: ' + name);
      if (curve.type === 'short') {
        var J = A.jmulAdd(new ref.BN(c.k1, 16), B, new ref.BN(c.k2, 16));
        if (!same(J.toP ? J.toP() : J, c.r)) throw new Error('off-curve jmulAdd mismatch: ' + name);
      }
      if (!c.on) nOff++;
    } else if (c.op === 'add') {
      var S = curve.point(c.p.x, c.p.y).add(curve.point(c.q.x, c.q.y));
      if (!same(S, c.r)) throw new Error('off-curve add mismatch: ' + name);
    } else {
      var key = { x: c.qx, y: c.qy };
      if (ec.verify(c.z, { r: c.r, s: c.s }, key) !== c.ok) throw new Error('off-curve verify mismatch: ' + name + ' ' + c.note);
      // ... and with the digest as BYTES, the form the patched EC#verify sends to the engine as one
      // (split) call: the off-curve status comes back beside the verdict and must be read after the
      // call has been collected (round 6: it was read before, and the reference's `true` was lost)
      if (c.z.length % 2 === 0 && ec.verify(Buffer.from(c.z, 'hex'), { r: c.r, s: c.s }, key) !== c.ok)
        throw new Error('off-curve verify mismatch (byte digest): ' + name + ' ' + c.note);
      items.push({ msg: Buffer.from(c.z, 'hex'), signature: { r: c.r, s: c.s }, key: key });
      wants.push(c.ok);
    }
    checked++;
  });
  if (eng.stats.offCurve - before < nOff) throw new Error('off-curve items did not reach the engine: ' + name);
  if (!ec) return;
  var got = eng.verifyMany(ec, items);
  for (var i = 0; i < items.length; i++)
    if (got[i] !== wants[i]) throw new Error('off-curve verifyMany mismatch: ' + name + ' #' + i);
  checked += items.length;
  pendingAsync.push(Promise.all(items.map(function(it) {
    return eng.verifyAsync(ec, it.msg, it.signature, it.key);
  })).then(function(res) {
    for (var i = 0; i < items.length; i++)
      if (res[i] !== wants[i]) throw new Error('off-curve verifyAsync mismatch: ' + name + ' #' + i);
    checked += items.length;
  }));
});
// api_forms.json (tools/api_forms.js): arguments that are objects of the library -- the patched
// library must answer, or throw, exactly what the unpatched reference did when the file was made
(function() {
  var forms = require('./api_forms');
  var byOp = {};
  load('api_forms.json').forEach(function(o, i) {
    var got = forms.run(elliptic, o);
    if (got !== o.want) throw new Error('api_forms #' + i + ' ' + JSON.stringify(o).slice(0, 300) + ': patched ' + got + ', reference ' + o.want);
    if (o.want[0] === 'e') thrown++;
    byOp[o.op] = (byOp[o.op] || 0) + 1;
    checked++;
  });
  if ((byOp['eddsa-verify'] || 0) < 200 || (byOp['sign-width'] || 0) < 300 || (byOp['foreign-sign'] || 0) < 25)
    throw new Error('api_forms.json is not the file tools/gen_golden.js writes: ' + JSON.stringify(byOp));
})();
// trusted_inputs.json (tools/trusted_inputs.js): precomputed tables that are not the point's
// multiples, endomorphism constants that are not the curve's, curves without a (complete) group law
// -- the reference computes with what it is given, and so must the patched library
(function() {
  var trusted = require('./trusted_inputs');
  var byOp = {};
  load('trusted_inputs.json').forEach(function(o, i) {
    var got = trusted.run(elliptic, o);
    if (got !== o.want) throw new Error('trusted_inputs #' + i + ' ' + JSON.stringify(o).slice(0, 300) + ': patched ' + got + ', reference ' + o.want);
    if (o.want[0] === 'e') thrown++;
    byOp[o.op] = (byOp[o.op] || 0) + 1;
    checked++;
  });
  if ((byOp.tables || 0) < 500 || (byOp['g-tables'] || 0) < 30 || (byOp.endo || 0) < 20 || (byOp.toy || 0) < 25 || (byOp['toy-endo'] || 0) < 6 || (byOp['foreign-red'] || 0) < 10 || (byOp.mutate || 0) < 130)
    throw new Error('trusted_inputs.json is not the file tools/gen_golden.js writes: ' + JSON.stringify(byOp));
})();
Promise.all(pendingAsync).then(function() {
  console.log(JSON.stringify({ ok: true, checked: checked, thrown: thrown, engine: eng.stats }));
  process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
}, function(e) { console.error(e.stack || e); process.exit(1); });
