'use strict';
// EDDSA#verify (lib/elliptic/eddsa/index.js:52-63) on edge encodings, patched library and batch API
// against the unpatched reference: public keys and R values whose y is 0, 1, 2, 18, p - 1 and the
// NON-CANONICAL forms p, p + 1, p + 2, p + 18 (the reference reduces y mod p in pointFromY), both
// sign bits (x = 0 with the sign bit set is where pointFromY throws), and S = 0, 1, 5, n - 1, n.
// Low-order points make some of these verify for any message; every verdict and every throw must
// be the reference's.
//   ELLGPU_LIB=<hostsim or real library> node tools/check_eddsa_edge_encodings.js
var loader = require('./ref_loader');
var ref = loader.load();
var plain = ref.elliptic, patched = loader.load().elliptic, BN = ref.BN;
var eng = require('../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB });
function run(f) { try { return { v: f() }; } catch (e) { return { e: String(e.message) }; } }
var a = new plain.eddsa('ed25519'), b = new patched.eddsa('ed25519');
var p = a.curve.p;
function enc(y, odd) { var bytes = y.toArray('le', 32); if (odd) bytes[31] |= 0x80; return bytes; }
var ys = [new BN(1), p.addn(1), new BN(0), p.clone(), p.subn(1), new BN(2), p.addn(2), new BN(18), p.addn(18),
  p.addn(19).maskn(255),
  // the order-8 points' y (and its negation): decodable, torsion
  new BN('05fc536d880238b13933c6d305acdfd5f098eff289f4c345b027b2c28f95e826', 16),
  p.sub(new BN('05fc536d880238b13933c6d305acdfd5f098eff289f4c345b027b2c28f95e826', 16))];
var msg = [1, 2, 3];
var n = 0, nTrue = 0, nThrow = 0;
ys.forEach(function(yA) { [false, true].forEach(function(oA) {
  ys.forEach(function(yR) { [false, true].forEach(function(oR) {
    [new BN(0), new BN(1), a.curve.n.subn(1), a.curve.n.clone(), new BN(5)].forEach(function(S) {
      var pub = enc(yA, oA), sig = enc(yR, oR).concat(S.toArray('le', 32));
      var w = run(function() { return a.verify(msg, sig, pub); });
      var g = run(function() { return b.verify(msg, sig, pub); });
      var r = eng.eddsaVerifyBatch([Buffer.from(msg)], Buffer.from(sig), Buffer.from(pub));
      var gb = r.err[0] ? { e: true } : { v: r.ok[0] === 1 };
      n++;
      if (w.v === true) nTrue++;
      if (w.e !== undefined) nThrow++;
      if (JSON.stringify(w) !== JSON.stringify(g) || (w.e !== undefined) !== (gb.e !== undefined) ||
          (w.e === undefined && w.v !== gb.v)) {
        console.log(JSON.stringify({ ok: false, yA: yA.toString(16), oA: oA, yR: yR.toString(16), oR: oR,
          S: S.toString(16), reference: w, patched: g, batch: gb }));
        process.exit(1);
      }
    });
  }); });
}); });
if (nTrue < 10 || nThrow < 100) { console.log(JSON.stringify({ ok: false, error: 'degenerate test set', accepted: nTrue, thrown: nThrow })); process.exit(1); }
console.log(JSON.stringify({ ok: true, cases: n, accepted: nTrue, thrown: nThrow, engine: eng.stats }));
process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
