'use strict';
// Exhaustive differential probe on toy curves: EVERY curve y^2 = x^3 + a x + b (or
// a x^2 + y^2 = 1 + d x^2 y^2) over F_p for a small prime p -- singular cubics and Edwards curves
// without a complete addition law included -- EVERY point on it, k = 0 .. 2p + 3 for Point#mul, a
// grid of Point#mulAdd / jmulAdd calls: the unpatched reference against a copy patched by
// install().  Results (canonical affine coordinates) and exception messages must be identical.
// A user-defined curve costs the engine a slot (16 per context), so a fresh pair of libraries is
// loaded for every dozen curves.  This is how round 5 found that singular cubics and incomplete
// Edwards curves must stay on the reference's own code (elliptic_amd/js/index.js customDomain).
//   ELLGPU_LIB=<hostsim or real library> node tools/probe_toy_curves.js <p> short|edwards
var loader = require('./ref_loader');
var ellgpu = require('../elliptic_amd/js');
var P = +process.argv[2] || 7, TYPE = process.argv[3] || 'short';

function canonP(v) {
  if (v.isInfinity() && v.curve.type !== 'edwards') return 'O';
  if (v.curve.type === 'edwards') { var q = v.curve.point(v.x, v.y, v.z, v.t); return '(' + q.getX().toString(16) + ',' + q.getY().toString(16) + ')'; }
  if (v.z !== undefined) v = v.toP();
  return '(' + v.getX().toString(16) + ',' + v.getY().toString(16) + ')';
}
function run(f) { try { return 'v:' + f(); } catch (e) { return 'e:' + String(e && e.message); } }
var specs = [];
for (var a = 0; a < P; a++) for (var b = 0; b < P; b++) specs.push([ a, b ]);
var total = 0, bad = 0, firstBad = {}, engineCalls = 0;

function work(L, a, b) {
  var BN = L.curves.secp256k1.curve.p.constructor;
  var c, pts = [], x, y;
  if (TYPE === 'short') {
    c = new L.curve.short({ p: P.toString(16), a: a.toString(16), b: b.toString(16) });
    for (x = 0; x < P; x++) for (y = 0; y < P; y++) if ((y * y - x * x * x - a * x - b) % P === 0) pts.push(c.point(new BN(x), new BN(y)));
  } else {
    if (a === 0 || b === 0 || a === b) return null;
    c = new L.curve.edwards({ p: P.toString(16), a: a.toString(16), c: '1', d: b.toString(16) });
    for (x = 0; x < P; x++) for (y = 0; y < P; y++) if ((a * x * x + y * y - 1 - b * x * x * y * y) % P === 0) pts.push(c.point(new BN(x), new BN(y)));
  }
  var out = [];
  pts.forEach(function(q, i) {
    for (var k = 0; k <= 2 * P + 3; k++) out.push(run(function() { return canonP(q.mul(new BN(k))); }));
    var r = pts[(i * 7 + 3) % pts.length];
    for (var k1 = 0; k1 < 4; k1++) for (var k2 = 0; k2 < 4; k2++)
      out.push(run(function() { return canonP(q.mulAdd(new BN(k1 + (i % 3)), r, new BN(k2 + 2 * P - 2))); }));
    if (TYPE === 'short') out.push(run(function() { return canonP(q.jmulAdd(new BN(3), r, new BN(P + 1))); }));
  });
  return out;
}
for (var s = 0; s < specs.length; s += 12) {
  var A = loader.load(), B = loader.load();
  var eng = ellgpu.install(B.elliptic, { libPath: process.env.ELLGPU_LIB });
  for (var t = s; t < Math.min(s + 12, specs.length); t++) {
    var ra = work(A.elliptic, specs[t][0], specs[t][1]), rb = work(B.elliptic, specs[t][0], specs[t][1]);
    if (!ra) continue;
    for (var i = 0; i < ra.length; i++) {
      total++;
      if (ra[i] !== rb[i]) { bad++; var key = specs[t].join(','); if (!firstBad[key]) firstBad[key] = [ i, ra[i], rb[i] ]; }
    }
  }
  engineCalls += eng.stats.gpuCalls;
  eng.uninstall();
  eng.close();                       // (the addon pins a context until it is closed)
}
console.log(JSON.stringify({ ok: bad === 0, type: TYPE, p: P, curves: specs.length, calls: total, mismatches: bad, engine_calls: engineCalls,
  first_mismatch_per_curve: firstBad }));
process.exit(bad === 0 ? 0 : 1);
