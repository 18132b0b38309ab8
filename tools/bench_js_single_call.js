'use strict';
// What ONE call of the reference's public API costs, unpatched and patched (install()), on the
// same host: EC#verify (ec/index.js:188-229), Point#mul (short.js:422-432), EC#sign
// (ec/index.js:109-172), and eng.verifyAsync for 1 and for 64 concurrent calls.  GPU box.
//   node tools/bench_js_single_call.js
var loader = require('./ref_loader');
var crypto = require('crypto');
var plain = loader.load().elliptic;
var patched = loader.load().elliptic;
// (ELLGPU_JS: another copy of the JavaScript layer, for A/B runs on one box)
var eng = require(process.env.ELLGPU_JS || '../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB });

function stats(ts) {
  ts.sort(function(a, b) { return a - b; });
  return { median_us: +(ts[ts.length >> 1] / 1e3).toFixed(1), best_us: +(ts[0] / 1e3).toFixed(1) };
}
function timeSync(fn, reps) {
  fn(); fn();
  var ts = [];
  for (var i = 0; i < reps; i++) {
    var t0 = process.hrtime.bigint();
    fn();
    ts.push(Number(process.hrtime.bigint() - t0));
  }
  return stats(ts);
}
function out(op, which, st, extra) {
  console.log(JSON.stringify(Object.assign({ op: op, library: which }, st, extra || {})));
}

var libs = [ [ 'reference (unpatched)', plain ], [ 'patched (install)', patched ] ];
var msg = crypto.createHash('sha256').update('single call').digest();
libs.forEach(function(l) {
  var ec = new l[1].ec('secp256k1');
  var kp = ec.genKeyPair({ entropy: crypto.createHash('sha512').update('k').digest() });
  var sig = kp.sign(msg);
  var der = sig.toDER('hex'), pub = kp.getPublic('hex');
  if (ec.verify(msg, der, pub, 'hex') !== true) throw new Error('verify');
  out('EC#verify (DER hex signature, hex key)', l[0], timeSync(function() { return ec.verify(msg, der, pub, 'hex'); }, 200));
  var key = ec.keyFromPublic(pub, 'hex');
  out('EC#verify (Signature object, KeyPair)', l[0], timeSync(function() { return ec.verify(msg, sig, key); }, 200));
  var P = kp.getPublic(), k = kp.getPrivate();
  out('Point#mul (variable base)', l[0], timeSync(function() { return P.mul(k).getX(); }, 200));
  out('Point#mul (fixed base: G.mul(k), KeyPair#getPublic)', l[0], timeSync(function() { return ec.g.mul(k).getX(); }, 200));
  var xh = P.getX().toString(16, 64), yOdd = P.getY().isOdd();
  out('ShortCurve#pointFromX', l[0], timeSync(function() { return ec.curve.pointFromX(xh, yOdd).getY(); }, 200));
  var pubc = kp.getPublic(true, 'hex');
  out('EC#verify (DER hex signature, COMPRESSED hex key)', l[0], timeSync(function() { return ec.verify(msg, der, pubc, 'hex'); }, 200));
  out('EC#sign', l[0], timeSync(function() { return ec.sign(msg, kp); }, 200));
  out('EC#recoverPubKey', l[0], timeSync(function() { return ec.recoverPubKey(msg, sig, sig.recoveryParam); }, 200));
  var Q = ec.genKeyPair({ entropy: crypto.createHash('sha512').update('q').digest() }).getPublic();
  out('Point#mulAdd (k1 G + k2 Q)', l[0], timeSync(function() { return ec.g.mulAdd(k, Q, sig.s).getX(); }, 200));
  // a curve without the endomorphism
  var e2 = new l[1].ec('p256');
  var kp2 = e2.genKeyPair({ entropy: crypto.createHash('sha512').update('k2').digest() });
  var der2 = kp2.sign(msg).toDER('hex'), pub2 = kp2.getPublic('hex');
  out('p256 EC#verify (DER hex signature, hex key)', l[0], timeSync(function() { return e2.verify(msg, der2, pub2, 'hex'); }, 100));
  out('p256 EC#sign', l[0], timeSync(function() { return e2.sign(msg, kp2); }, 100));
  // the wide NIST curves (one item per lane also for a lone call: their fields do not fit a 16-lane row)
  [ 'p384', 'p521' ].forEach(function(name) {
    var e3 = new l[1].ec(name);
    var kp3 = e3.genKeyPair({ entropy: crypto.createHash('sha512').update('k3' + name).digest() });
    var m3 = crypto.createHash(name === 'p384' ? 'sha384' : 'sha512').update('single call').digest();
    var der3 = kp3.sign(m3).toDER('hex'), pub3 = kp3.getPublic('hex');
    if (e3.verify(m3, der3, pub3, 'hex') !== true) throw new Error('verify ' + name);
    out(name + ' EC#verify (DER hex signature, hex key)', l[0], timeSync(function() { return e3.verify(m3, der3, pub3, 'hex'); }, 40));
    out(name + ' EC#sign', l[0], timeSync(function() { return e3.sign(m3, kp3); }, 40));
    var P3 = kp3.getPublic(), k3 = kp3.getPrivate();
    out(name + ' Point#mul (variable base)', l[0], timeSync(function() { return P3.mul(k3).getX(); }, 40));
  });
  var ed = new l[1].eddsa('ed25519');
  var ek = ed.keyFromSecret(crypto.createHash('sha256').update('ed').digest());
  var esig = ek.sign(msg).toHex(), epub = ek.getPublic('hex');
  out('EDDSA#verify', l[0], timeSync(function() { return ed.verify(msg, esig, epub); }, 100));
  out('EDDSA#sign', l[0], timeSync(function() { return ed.sign(msg, ek).toHex(); }, 100));
  // the other two curve families' own Point#mul
  var eA = ek.pub(), ekk = kp.getPrivate();
  out('ed25519 Point#mul', l[0], timeSync(function() { return eA.mul(ekk).getX(); }, 100));
  var c25 = new l[1].ec('curve25519');
  var mk = c25.keyFromPrivate(crypto.createHash('sha256').update('m').digest());
  var mP = c25.keyFromPrivate(crypto.createHash('sha256').update('n').digest()).getPublic();
  var mkk = mk.getPrivate();
  out('curve25519 Point#mul (x only)', l[0], timeSync(function() { return mP.mul(mkk).getX(); }, 100));
  // (KeyPair#derive = MontCurve#validate -- a square root in JavaScript, not patched -- + this mul)
  out('curve25519 KeyPair#derive', l[0], timeSync(function() { return mk.derive(mP); }, 100));
});

// the engine's own asynchronous single call, alone and 64 at a time (one launch)
var ecq = new patched.ec('secp256k1');
var kps = [], jobs = [];
for (var i = 0; i < 64; i++) {
  var kp = ecq.genKeyPair({ entropy: crypto.createHash('sha512').update('a' + i).digest() });
  var m = crypto.createHash('sha256').update('m' + i).digest();
  jobs.push({ msg: m, sig: kp.sign(m).toDER('hex'), key: kp.getPublic('hex') });
}
function timeAsync(n, reps, done) {
  var ts = [], i = 0;
  function one() {
    if (i++ === reps) return done(stats(ts));
    var t0 = process.hrtime.bigint();
    Promise.all(jobs.slice(0, n).map(function(j) { return eng.verifyAsync(ecq, j.msg, j.sig, j.key, 'hex'); }))
      .then(function(v) {
        if (v.indexOf(false) >= 0) throw new Error('verifyAsync');
        ts.push(Number(process.hrtime.bigint() - t0));
        one();
      });
  }
  one();
}
timeAsync(1, 200, function(s1) {
  out('eng.verifyAsync x 1', 'patched (install)', s1);
  timeAsync(64, 100, function(s64) {
    out('eng.verifyAsync x 64 concurrent (one launch)', 'patched (install)', s64, { per_verify_us: +(s64.median_us / 64).toFixed(1) });
    process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
  });
});
