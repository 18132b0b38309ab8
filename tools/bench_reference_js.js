'use strict';
// The reference's own pure-JS CPU path, timed.  Runs ONLY in the build container (it needs
// /root/reference; the GPU box has no copy of the reference, so bench.py quotes this file's
// output beside its own C-port baseline).  Mirrors benchmarks/index.js:50-55,106-109 (one fixed
// message / key / signature, ec.verify in a loop) and adds a seeded random set, G*k and P*k.
//
//   node tools/bench_reference_js.js [seconds] > profiles/r01_reference_js_cpu.json
var os = require('os');
var crypto = require('crypto');
var ref = require('./ref_loader').load();
var elliptic = ref.elliptic;
var secs = parseFloat(process.argv[2] || '3');

function rate(fn) {
  fn();
  var n = 0, t0 = process.hrtime.bigint(), dt;
  do { fn(); n++; dt = Number(process.hrtime.bigint() - t0) / 1e9; } while (dt < secs);
  return n / dt;
}
var ec = new elliptic.ec('secp256k1');
var msg = crypto.createHash('sha256').update('big benchmark against elliptic').digest();
var key = ec.keyFromPrivate(crypto.createHash('sha256').update('ellgpu-bench-v1:1').digest());
var sig = ec.sign(msg, key);
var pub = ec.keyFromPublic(key.getPublic());
var out = {
  what: 'indutny/elliptic ' + elliptic.version + ' (dist/elliptic.js, bn.js 4.11.9), single thread',
  node: process.version, cpu: os.cpus()[0].model, logical_cpus: os.cpus().length, seconds_per_row: secs,
  where: 'build container (the GPU box holds no copy of the reference)',
};
out.verify_fixed_per_s = rate(function() { if (!ec.verify(msg, sig, pub)) throw new Error('verify'); });
// seeded random set: 256 keys / messages / signatures, cycled
var set = [];
for (var i = 0; i < 256; i++) {
  var d = crypto.createHash('sha256').update('ellgpu-bench-v1:key:' + i).digest();
  var k = ec.keyFromPrivate(d);
  var m = crypto.createHash('sha256').update('ellgpu-bench-v1:msg:' + i).digest();
  set.push({ m: m, s: ec.sign(m, k), p: ec.keyFromPublic(k.getPublic()), d: k.getPrivate() });
}
var j = 0;
out.verify_random_per_s = rate(function() { var c = set[j++ & 255]; if (!ec.verify(c.m, c.s, c.p)) throw new Error('verify'); });
out.sign_per_s = rate(function() { var c = set[j++ & 255]; ec.sign(c.m, c.d); });
out.mul_fixed_per_s = rate(function() { ec.g.mul(set[j++ & 255].d); });
out.mul_var_per_s = rate(function() { var c = set[j++ & 255]; c.p.getPublic().mul(set[(j + 7) & 255].d); });
var ed = new elliptic.eddsa('ed25519');
var ek = ed.keyFromSecret(crypto.createHash('sha256').update('ellgpu-bench-v1:ed').digest());
var es = ek.sign(msg);
out.eddsa_verify_per_s = rate(function() { if (!ek.verify(msg, es)) throw new Error('eddsa'); });
var e384 = new elliptic.ec('p384');
var k384 = e384.keyFromPrivate(crypto.createHash('sha384').update('ellgpu-bench-v1:p384').digest());
var q384 = k384.getPublic();
out.p384_mul_var_per_s = rate(function() { q384.mul(set[j++ & 255].d); });
console.log(JSON.stringify(out));
