'use strict';
// Golden vectors for USER-DEFINED short curves (`new elliptic.curve.short({p, a, b, ...})` with
// parameters that are no preset): Point#mul, mulAdd / jmulAdd, Point#add / dbl of the reference
// itself.  Runs only where the reference is present (see tools/ref_loader.js); all randomness is
// SHA-256 counter mode over a fixed seed, so a rerun reproduces tests/golden/custom_short.json
// byte for byte.
//
//   node tools/gen_golden_custom.js [outdir]
//
// Curves: the two custom curves of the reference's own test/curve-test.js:9-22, 114-127 (the
// 5-bit example curve and secp112r1), brainpoolP256r1 (RFC 5639: a 256-bit prime without special
// form and an arbitrary a), secp192k1 (SEC 2: a = 0 without being secp256k1) and a twist-like
// 255-bit curve over 2^255 - 19 whose base point the script finds with pointFromX.

var fs = require('fs');
var path = require('path');
var crypto = require('crypto');
var ref = require('./ref_loader').load();
var elliptic = ref.elliptic;
var BN = ref.BN;

var OUT = process.argv[2] || path.join(__dirname, '..', 'tests', 'golden');

function Prng(seed) { this.seed = seed; this.ctr = 0; }
Prng.prototype.bytes = function(n) {
  var out = [];
  while (out.length < n) {
    var h = crypto.createHash('sha256').update(this.seed + ':' + (this.ctr++)).digest();
    for (var i = 0; i < h.length && out.length < n; i++) out.push(h[i]);
  }
  return Buffer.from(out);
};
Prng.prototype.bits = function(b) { return new BN(this.bytes(Math.ceil(b / 8))).maskn(b); };

function hex32(bn) { return bn.toString(16, 64); }
function aff(p) {
  if (p.isInfinity()) return { inf: true };
  if (p.z !== undefined) p = p.toP();
  return { x: hex32(p.getX()), y: hex32(p.getY()) };
}

var CURVES = [
  { name: 'example_p29', p: '1d', a: '4', b: '14', g: ['18', '16'] },
  { name: 'secp112r1', p: 'db7c2abf62e35e668076bead208b', a: 'db7c2abf62e35e668076bead2088',
    b: '659ef8ba043916eede8911702b22', g: ['09487239995a5ee76b55f9c2f098', 'a89ce5af8724c0a23e0e0ff77500'],
    n: 'db7c2abf62e35e7628dfac6561c5' },
  { name: 'brainpoolP256r1',
    p: 'a9fb57dba1eea9bc3e660a909d838d726e3bf623d52620282013481d1f6e5377',
    a: '7d5a0975fc2c3057eef67530417affe7fb8055c126dc5c6ce94a4b44f330b5d9',
    b: '26dc5c6ce94a4b44f330b5d9bbd77cbf958416295cf7e1ce6bccdc18ff8c07b6',
    g: ['8bd2aeb9cb7e57cb2c4b482ffc81b7afb9de27e1e3bd23c23a4453bd9ace3262',
      '547ef835c3dac4fd97f8461a14611dc9c27745132ded8e545c1d54c72f046997'],
    n: 'a9fb57dba1eea9bc3e660a909d838d718c397aa3b561a6f7901e0e82974856a7' },
  { name: 'secp192k1', p: 'fffffffffffffffffffffffffffffffffffffffeffffee37', a: '0', b: '3',
    g: ['db4ff10ec057e9ae26b07d0280b7f4341da5d1b1eae06c7d', '9b2f2f6d9c5628a7844163d015be86344082aa88d95e2f9d'],
    n: 'fffffffffffffffffffffffe26f2fc170f69466a74defd8d' },
  { name: 'w25519_like', p: '7fffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffed',
    a: '2aaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaaa984914a144',
    b: '7b425ed097b425ed097b425ed097b425ed097b425ed097b4260b5e9c7710c864' },
];

function build(spec) {
  var conf = { p: spec.p, a: spec.a, b: spec.b };
  if (spec.n) conf.n = spec.n;
  var curve = new elliptic.curve.short(conf);
  var g;
  if (spec.g) g = curve.point(spec.g[0], spec.g[1]);
  else {
    for (var x = 1; ; x++) {
      try { g = curve.pointFromX(new BN(x), false); break; } catch (e) { /* next x */ }
    }
  }
  if (!curve.validate(g)) throw new Error(spec.name + ': base point not on the curve');
  if (spec.n && !g.mul(new BN(spec.n, 16)).isInfinity()) throw new Error(spec.name + ': n * G != O');
  return { curve: curve, g: g };
}

function gen(spec) {
  var b = build(spec);
  var curve = b.curve, G = b.g;
  var rng = new Prng('ellgpu-golden-v1:custom:' + spec.name);
  var n = spec.n ? new BN(spec.n, 16) : null;
  var cases = [];
  function rec(op, o) { o.op = op; cases.push(o); }
  function pt(p) { return { x: hex32(p.getX()), y: hex32(p.getY()) }; }
  function randPoint() {
    var k = rng.bits(curve.p.bitLength() + 8);
    var p = G.mul(k);
    return p.isInfinity() ? G : p;
  }
  var edge = [new BN(0), new BN(1), new BN(2), new BN(3), new BN(15), new BN(16), new BN(17),
    new BN(1).ushln(128), new BN(1).ushln(255), new BN(1).ushln(256).subn(1)];
  if (n) edge.push(n.subn(1), n.clone(), n.addn(1), n.subn(2), n.ushrn(1), n.ushrn(1).addn(1));
  var P0 = randPoint();
  edge.forEach(function(k) {
    rec('mul', { k: hex32(k), p: pt(P0), r: aff(P0.mul(k)) });
    rec('mul', { k: hex32(k), p: pt(G), r: aff(G.mul(k)) });
  });
  var i;
  for (i = 0; i < 24; i++) {
    var P = randPoint();
    var k = (i % 3 === 2) ? rng.bits(256) : rng.bits(curve.p.bitLength());
    rec('mul', { k: hex32(k), p: pt(P), r: aff(P.mul(k)) });
  }
  var one = new BN(1);
  var special = [
    [G, new BN(5), P0, new BN(7)], [P0, one, P0.neg(), one], [P0, one, P0, one],
    [G, new BN(0), P0, new BN(0)], [G, new BN(0), P0, new BN(9)], [G, new BN(9), P0, new BN(0)],
    [G, new BN(2), G.neg(), new BN(2)], [P0, new BN(3), P0, new BN(5)],
  ];
  if (n) special.push([G, n.subn(1), G, one], [G, n.subn(2), G, new BN(2)]);
  special.forEach(function(s) {
    rec('muladd', { k1: hex32(s[1]), p1: pt(s[0]), k2: hex32(s[3]), p2: pt(s[2]),
      r: aff(s[0].mulAdd(s[1], s[2], s[3])), rj: aff(s[0].jmulAdd(s[1], s[2], s[3])) });
  });
  for (i = 0; i < 12; i++) {
    var A = randPoint(), B = randPoint();
    var k1 = rng.bits(curve.p.bitLength()), k2 = rng.bits(curve.p.bitLength());
    rec('muladd', { k1: hex32(k1), p1: pt(A), k2: hex32(k2), p2: pt(B),
      r: aff(A.mulAdd(k1, B, k2)), rj: aff(A.jmulAdd(k1, B, k2)) });
  }
  // Point#add (short.js:365-392) incl. P + P (-> dbl), P + -P, O on either side
  var O = curve.point(null, null);
  var Q0 = randPoint();
  var adds = [[P0, Q0], [P0, P0], [P0, P0.neg()], [O, P0], [P0, O], [O, O], [G, G]];
  for (i = 0; i < 8; i++) adds.push([randPoint(), randPoint()]);
  adds.forEach(function(s) {
    function side(p) { return p.isInfinity() ? { inf: true } : pt(p); }
    rec('add', { p1: side(s[0]), p2: side(s[1]), r: aff(s[0].add(s[1])) });
  });
  // JPoint#dblp (short.js:605-654: the generic-a repeated doubling) through its affine results
  [1, 2, 5].forEach(function(pow) {
    rec('dblp', { p: pt(P0), pow: pow, r: aff(P0.toJ().dblp(pow)) });
  });
  return { name: spec.name, p: hex32(curve.p), a: hex32(curve.a.fromRed()), b: hex32(curve.b.fromRed()),
    n: n ? hex32(n) : null, g: pt(G), cases: cases };
}

var out = CURVES.map(gen);
var file = path.join(OUT, 'custom_short.json');
fs.writeFileSync(file, JSON.stringify(out, null, 1) + '\n');
console.log('wrote', file, out.map(function(c) { return c.name + ':' + c.cases.length; }).join(' '));

// ---- user-defined (twisted) Edwards curves, c = 1: custom_edwards.json ---------------------------
// Curve1174 (p = 2^251 - 9, a = 1, d = -1174) and E-222 (p = 2^222 - 117, a = 1, d = 160102):
// a = 1, so the reference runs _projDbl / _projAdd in their untwisted branches; two curves over
// 2^255 - 19: a = 4 (twisted, projective branch with _mulA) and a = -1 with a d that is not
// ed25519's (extended coordinates).  d is a non-square on all four (checked below), so the
// addition laws are complete.  Base points come from pointFromY.
var ED = [
  { name: 'curve1174', p: new BN(1).ushln(251).subn(9), a: new BN(1), d: new BN(1).ushln(251).subn(9).subn(1174) },
  { name: 'e222', p: new BN(1).ushln(222).subn(117), a: new BN(1), d: new BN(160102) },
  { name: 'twisted_a4', p: new BN(1).ushln(255).subn(19), a: new BN(4), d: null },
  { name: 'twisted_am1', p: new BN(1).ushln(255).subn(19), a: new BN(1).ushln(255).subn(20), d: null, aconf: '-1' },
];
function legendre(x, p) {
  var red = BN.red(p);
  return x.toRed(red).redPow(p.subn(1).ushrn(1)).fromRed().cmpn(1) === 0 ? 1 : -1;
}
function genEd(spec) {
  var d = spec.d;
  if (!d) for (d = new BN(121666); legendre(d, spec.p) !== -1; d = d.addn(1)) { /* next */ }
  if (legendre(d, spec.p) !== -1) throw new Error(spec.name + ': d is a square');
  var curve = new elliptic.curve.edwards({ p: spec.p.toString(16), a: spec.aconf || spec.a.toString(16), c: '1',
    d: d.toString(16) });
  var G = null;
  for (var y = 2; !G; y++) {
    try { G = curve.pointFromY(new BN(y), false); } catch (e) { G = null; }
    if (G && G.getX().isZero()) G = null;
  }
  if (!curve.validate(G)) throw new Error(spec.name + ': base point not on the curve');
  var rng = new Prng('ellgpu-golden-v1:custom-edwards:' + spec.name);
  var cases = [];
  function pt(p) { var q = curve.point(p.x, p.y, p.z, p.t); return { x: hex32(q.getX()), y: hex32(q.getY()) }; }
  function randPoint() { return G.mul(rng.bits(spec.p.bitLength() + 8)); }
  var P0 = randPoint();
  var edge = [new BN(0), new BN(1), new BN(2), new BN(3), new BN(7), new BN(8), new BN(9), new BN(15), new BN(16),
    new BN(17), new BN(1).ushln(128), new BN(1).ushln(255), new BN(1).ushln(256).subn(1)];
  edge.forEach(function(k) {
    cases.push({ op: 'mul', k: hex32(k), p: pt(P0), r: pt(P0.mul(k)) });
    cases.push({ op: 'mul', k: hex32(k), p: pt(G), r: pt(G.mul(k)) });
  });
  var i;
  for (i = 0; i < 24; i++) {
    var P = randPoint();
    var k = (i % 3 === 2) ? rng.bits(256) : rng.bits(spec.p.bitLength());
    cases.push({ op: 'mul', k: hex32(k), p: pt(P), r: pt(P.mul(k)) });
  }
  var O = curve.point(null, null, null);
  var Q0 = randPoint();
  var adds = [[P0, Q0], [P0, P0], [P0, P0.neg()], [O, P0], [P0, O], [O, O], [G, G]];
  for (i = 0; i < 10; i++) adds.push([randPoint(), randPoint()]);
  adds.forEach(function(s) {
    cases.push({ op: 'add', p1: pt(s[0]), p2: pt(s[1]), r: pt(s[0].add(s[1])) });
  });
  [P0, G, Q0].forEach(function(p) { cases.push({ op: 'dbl', p: pt(p), r: pt(p.dbl()) }); });
  return { name: spec.name, p: hex32(curve.p), a: hex32(curve.a.fromRed()), d: hex32(curve.d.fromRed()),
    extended: curve.extended, twisted: curve.twisted, g: pt(G), cases: cases };
}
var outE = ED.map(genEd);
file = path.join(OUT, 'custom_edwards.json');
fs.writeFileSync(file, JSON.stringify(outE, null, 1) + '\n');
console.log('wrote', file, outE.map(function(c) { return c.name + ':' + c.cases.length + (c.extended ? ':ext' : ':proj'); }).join(' '));
