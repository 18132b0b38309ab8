'use strict';
// Throughput of the batch API through the N-API addon (what a Node caller gets, PCIe and
// Buffer allocation included).  GPU box.   node elliptic_amd/js/bench.js [n]
var crypto = require('crypto');
var path = require('path');
var Engine = require('./index.js').Engine;
var eng = new Engine({ libPath: process.argv[3] });
var n = parseInt(process.argv[2] || '1048576', 10);
function rnd(bytes) {
  var out = Buffer.alloc(bytes);
  for (var o = 0; o < bytes; o += 1 << 20) crypto.randomBytes(Math.min(1 << 20, bytes - o)).copy(out, o);
  return out;
}
var k = rnd(n * 32), d = rnd(n * 32);
function timed(name, fn, reps) {
  fn();
  var best = Infinity, sum = 0;
  for (var i = 0; i < reps; i++) {
    var t0 = process.hrtime.bigint();
    fn();
    var dt = Number(process.hrtime.bigint() - t0) / 1e9;
    best = Math.min(best, dt); sum += dt;
  }
  console.log(JSON.stringify({ op: name, n: n, items_per_s_best: n / best, items_per_s_mean: n * reps / sum }));
}
var pts = eng.mulBatch('secp256k1', d, null).xy;
timed('mulBatch fixed-base (32 B in, 65 B out per item)', function() { return eng.mulBatch('secp256k1', d, null); }, 5);
timed('mulBatch variable-base (96 B in, 65 B out per item)', function() { return eng.mulBatch('secp256k1', k, pts); }, 5);
var keep = { xy: Buffer.alloc(n * 64), inf: Buffer.alloc(n) };
timed('mulBatch fixed-base, caller-owned result Buffers', function() { return eng.mulBatch('secp256k1', d, null, keep); }, 5);
timed('mulBatch variable-base, caller-owned result Buffers', function() { return eng.mulBatch('secp256k1', k, pts, keep); }, 5);
var sig = eng.ecdsaSignDetBatch('secp256k1', { hashes: k, hashLen: 32, priv: d, canonical: true });
timed('ecdsaSignDetBatch (64 B in, 66 B out per item)', function() {
  return eng.ecdsaSignDetBatch('secp256k1', { hashes: k, hashLen: 32, priv: d, canonical: true });
}, 5);
timed('ecdsaVerifyBatch (160 B in, 1 B out per item)', function() {
  return eng.ecdsaVerifyBatch('secp256k1', { hashes: k, hashLen: 32, r: sig.r, s: sig.s, pub: pts });
}, 5);
timed('ecdsaRecoverBatch (97 B in, 65 B out per item)', function() {
  return eng.ecdsaRecoverBatch('secp256k1', { hashes: k, hashLen: 32, r: sig.r, s: sig.s, recid: sig.recid });
}, 5);
