'use strict';
// A walk over the PUBLIC API with one set of inputs in every form the reference accepts -- keys as
// points, KeyPairs, hex / array / Buffer / Uint8Array encodings (uncompressed, compressed, hybrid,
// truncated), {x, y} objects; signatures as objects, DER in every container, high-s, r + n, broken
// DER; messages as arrays, Buffers, hex strings, BNs, numbers, empty, array-likes; every recovery
// id; pointFromX / decodePoint forms; EdDSA keys, signatures, secrets and messages likewise -- on
// an unpatched copy of the reference and on a copy patched by install().  Every result (canonical
// rendering) and every exception message must be identical.
//   ELLGPU_LIB=<hostsim or real library> node tools/probe_api_walk.js
var loader = require('./ref_loader');
var crypto = require('crypto');
var A = loader.load(), B = loader.load();
var plain = A.elliptic, patched = B.elliptic;
var eng = require('../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB });
function canon(v) {
  if (v === null || v === undefined || typeof v === 'boolean' || typeof v === 'number' || typeof v === 'string') return String(v);
  if (Array.isArray(v)) return '[' + v.map(canon).join(',') + ']';
  if (v.r && v.s && v.recoveryParam !== undefined) return 'sig(' + v.r.toString(16) + ',' + v.s.toString(16) + ',' + v.recoveryParam + ')';
  if (typeof v.isInfinity === 'function') {
    if (v.isInfinity() && v.curve.type !== 'edwards') return 'O';
    if (typeof v.toP === 'function' && v.z !== undefined && v.curve.type === 'short') v = v.toP();
    if (v.curve.type === 'mont') return 'x=' + v.getX().toString(16);
    var q = v.curve.type === 'edwards' ? v.curve.point(v.x, v.y, v.z, v.t) : v;
    return '(' + q.getX().toString(16) + ',' + q.getY().toString(16) + ')';
  }
  if (v.constructor && v.constructor.name === 'BN') return 'bn' + v.toString(16);
  if (typeof v.toHex === 'function') return 'hex' + v.toHex();
  return JSON.stringify(v);
}
function run(f) { try { return 'v:' + canon(f()); } catch (e) { return 'e:' + String(e && e.message); } }
var H = { run: run };
var groups = 0, calls = 0, thrown = 0, failures = [];
function both(name, f) {
  var a = f(plain), b = f(patched);
  groups++;
  for (var i = 0; i < a.length; i++) {
    calls++;
    if (a[i][0] === 'e') thrown++;
    if (a[i] !== b[i]) failures.push({ group: name, index: i, reference: a[i].slice(0, 200), patched: b[i].slice(0, 200) });
  }
}
function rb(s, n) { var o = []; var c = 0; while (o.length < n) { var h = crypto.createHash('sha256').update(s + ':' + (c++)).digest(); for (var i = 0; i < 32 && o.length < n; i++) o.push(h[i]); } return o; }
['secp256k1', 'p256', 'p224', 'p521'].forEach(function(cn) {
  both('api walk ' + cn, function(L) {
    var ec = new L.ec(cn), BN = ec.n.constructor, out = [];
    function t(f) { out.push(H.run(f)); }
    var kp = ec.genKeyPair({ entropy: rb('e' + cn, 64) });
    var kp2 = ec.keyFromPrivate(rb('p' + cn, ec.n.byteLength() + 3));            // wider than n: reduced
    var msg = rb('m', 32), pub = kp.getPublic();
    t(function() { return kp.getPublic('hex'); }); t(function() { return kp.getPublic(true, 'hex'); });
    t(function() { return kp2.getPublic(true, 'hex'); }); t(function() { return JSON.stringify(kp.validate()); });
    t(function() { return kp.derive(kp2.getPublic()).toString(16); }); t(function() { return kp2.derive(pub).toString(16); });
    var sig = kp.sign(msg), sigc = kp.sign(msg, { canonical: true });
    t(function() { return sig; }); t(function() { return sigc; }); t(function() { return sig.toDER('hex'); });
    // key forms
    var forms = [ pub, kp, kp.getPublic('hex'), kp.getPublic(true, 'hex'), kp.getPublic('array'), kp.getPublic(true, 'array'),
      Buffer.from(kp.getPublic('array')), new Uint8Array(kp.getPublic(true, 'array')), { x: pub.getX().toString(16), y: pub.getY().toString(16) },
      { x: pub.getX(), y: pub.getY() }, { x: pub.getX().toArray(), y: pub.getY().toArray() }, { x: pub.getX().toString(16) }, '06' + kp.getPublic('hex').slice(2), '07' + kp.getPublic('hex').slice(2),
      '04' + kp.getPublic('hex').slice(2, 20), '', null, undefined, 5 ];
    // signature forms
    var sforms = [ sig, sig.toDER(), sig.toDER('hex'), Buffer.from(sig.toDER()), new Uint8Array(sig.toDER()), { r: sig.r, s: sig.s }, { r: sig.r.toString(16), s: sig.s.toString(16) },
      { r: sig.r.toArray(), s: sig.s.toArray() }, { r: sig.r.toString(16) }, { r: sig.r, s: ec.n.sub(sig.s) }, { r: sig.r.add(ec.n), s: sig.s }, { r: 0, s: 1 }, { r: '00' + sig.r.toString(16), s: sig.s },
      sig.toDER('hex') + '00', '30' + sig.toDER('hex').slice(2), [], '', null, 7 ];
    forms.forEach(function(k, i) { t(function() { return ec.verify(msg, sig, k, typeof k === 'string' ? 'hex' : undefined); }); });
    sforms.forEach(function(s, i) { t(function() { return ec.verify(msg, s, pub); }); t(function() { return kp.verify(msg, s); }); });
    // message forms
    [ msg, Buffer.from(msg), new Uint8Array(msg), Buffer.from(msg).toString('hex'), new BN(msg), msg.concat([1, 2, 3]), msg.slice(0, 5), [], '', 0, 12345, null, { length: 2, 0: 1, 1: 2 } ].forEach(function(m) {
      t(function() { return ec.verify(m, sig, pub); }); t(function() { return ec.sign(m, kp); });
      t(function() { var s2 = ec.sign(m, kp); return ec.recoverPubKey(m, s2, s2.recoveryParam); });
    });
    [ 0, 1, 2, 3, 4, -1, '1', null ].forEach(function(j) { t(function() { return ec.recoverPubKey(msg, sig, j); }); });
    t(function() { return ec.getKeyRecoveryParam(msg, sig, pub); }); t(function() { return ec.getKeyRecoveryParam(msg, sig.toDER('hex'), pub, 'hex'); });
    // decodePoint / pointFromX
    t(function() { return ec.curve.decodePoint(kp.getPublic(true, 'hex'), 'hex'); }); t(function() { return ec.curve.pointFromX(pub.getX(), true); });
    t(function() { return ec.curve.pointFromX(pub.getX().toString(16), 0); }); t(function() { return ec.curve.pointFromX(pub.getX().toArray(), 1); });
    t(function() { return ec.curve.pointFromX(new BN(5), false); }); t(function() { return ec.curve.pointFromX(ec.curve.p.addn(5), false); }); t(function() { return ec.curve.pointFromX(new BN(-5), false); });
    t(function() { return ec.curve.pointFromX(pub.getX().toRed(ec.curve.red), true); });
    return out;
  });
});
both('eddsa walk', function(L) {
  var ed = new L.eddsa('ed25519'), out = []; function t(f) { out.push(H.run(f)); }
  var sec = rb('s', 32), key = ed.keyFromSecret(sec), msg = rb('m', 40);
  var sig = key.sign(msg);
  t(function() { return sig.toHex(); }); t(function() { return key.getPublic('hex'); }); t(function() { return key.verify(msg, sig); });
  [ sig, sig.toHex(), sig.toBytes(), Buffer.from(sig.toBytes()), sig.toHex().slice(2), sig.toHex() + '00', { R: sig.R(), S: sig.S() }, { R: sig.Rencoded(), S: sig.Sencoded() }, null, 5, '' ].forEach(function(s) {
    [ key, key.getPublic(), key.getPublic('hex'), Buffer.from(key.getPublic()), ed.keyFromPublic(key.getPublic()), key.pub(), key.getPublic('hex').slice(2), null, '' ].forEach(function(k) {
      t(function() { return ed.verify(msg, s, k); });
    });
  });
  [ msg, Buffer.from(msg), Buffer.from(msg).toString('hex'), new Uint8Array(msg), [], '', 'zz', null, 5 ].forEach(function(m) {
    t(function() { return ed.sign(m, sec).toHex(); }); t(function() { return ed.sign(m, key).toHex(); }); t(function() { return ed.verify(m, sig, key); });
  });
  [ sec, Buffer.from(sec), Buffer.from(sec).toString('hex'), sec.slice(1), sec.concat([1]), [], null ].forEach(function(s) { t(function() { return ed.sign(msg, s).toHex(); }); t(function() { return ed.keyFromSecret(s).getPublic('hex'); }); });
  return out;
});
console.log(JSON.stringify({ ok: failures.length === 0, groups: groups, calls: calls, reference_threw: thrown, failures: failures.slice(0, 5), engine: eng.stats }));
process.exit(failures.length ? 1 : 0);
