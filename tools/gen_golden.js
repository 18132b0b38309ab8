'use strict';
// Golden-vector generator.  Runs ONLY in the build container, where
// /root/reference exists and Node is present.  It drives the reference
// implementation itself (dist/elliptic.js = lib/ + vendored bn.js 4.11.9, see
// SURVEY.md §8c) and writes small JSON fixtures to tests/golden/:
//
//   curves.json             preset parameters as the reference holds them
//   mul_<curve>.json        seeded + edge cases for Point.mul / mulAdd
//   verify_<curve>.json     ECDSA verify tuples (valid + corrupted)
//   offcurve_<curve>.json   points that are not on the curve: what the reference answers
//   trusted_inputs.json     points with precomputed tables of their own (right and wrong), curves with
//                           given endomorphism constants, singular / incomplete toy curves, the
//                           private ladders with Jacobian operands (tools/trusted_inputs.js)
//   api_forms.json          calls whose arguments are library OBJECTS (points, KeyPairs, Signatures):
//                           EDDSA#verify over every key form x signature form, EC#sign / verify over
//                           digest widths x options.msgBitLength, KeyPairs of another EC instance
//   captured_<curve>.json   every call the reference's OWN mocha suite makes
//                           into the hot path (Point.mul / mulAdd / jmulAdd /
//                           ec.verify), captured at the prototype boundary
//                           (tools/run_ref_tests.js instrumentation)
//
// All randomness of the seeded fixtures is SHA-256 counter mode over a fixed seed, so
// rerunning this script reproduces them byte for byte; captured_*.json depends on the
// reference suite's own key generation and differs from run to run (every capture is
// still an input/output pair of the reference).
//
//   node tools/gen_golden.js [outdir]

var fs = require('fs');
var path = require('path');
var crypto = require('crypto');
var ref = require('./ref_loader').load();
var elliptic = ref.elliptic;
var BN = ref.BN;

var OUT = process.argv[2] || path.join(__dirname, '..', 'tests', 'golden');

// ---------------------------------------------------------------- PRNG ----
function Prng(seed) { this.seed = seed; this.ctr = 0; }
Prng.prototype.bytes = function(n) {
  var out = [];
  while (out.length < n) {
    var h = crypto.createHash('sha256')
      .update(this.seed + ':' + (this.ctr++)).digest();
    for (var i = 0; i < h.length && out.length < n; i++) out.push(h[i]);
  }
  return Buffer.from(out);
};
Prng.prototype.below = function(n) {          // uniform-ish in [0, n)
  var len = n.byteLength() + 8;
  return new BN(this.bytes(len)).umod(n);
};
Prng.prototype.bits = function(b) {
  var x = new BN(this.bytes(Math.ceil(b / 8)));
  return x.maskn(b);
};

function hex(bn, bytes) { return bn.toString(16, bytes * 2); }

var SHORT = ['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521'];
var COUNTS = { secp256k1: 96, p192: 24, p224: 24, p256: 32, p384: 24,
  p521: 12, ed25519: 48, curve25519: 32 };

function flen(c) { return c.p.byteLength(); }

function affine(curve, p) {
  // -> {inf:true} or {x,y} hex, never mutating p
  if (p.isInfinity()) return { inf: true };
  var L = flen(curve);
  if (curve.type === 'short') {
    if (p.z !== undefined) p = p.toP();
    return { x: hex(p.getX(), L), y: hex(p.getY(), L) };
  }
  if (curve.type === 'edwards') {
    var q = curve.point(p.x, p.y, p.z, p.t);
    return { x: hex(q.getX(), L), y: hex(q.getY(), L) };
  }
  var m = curve.point(p.x, p.z);
  return { x: hex(m.getX(), L) };
}

// ------------------------------------------------------------- curves.json
function dumpCurves() {
  var out = {};
  SHORT.concat(['ed25519', 'curve25519']).forEach(function(name) {
    var pc = elliptic.curves[name];
    var c = pc.curve;
    var L = flen(c);
    var o = { type: c.type, p: hex(c.p, L), bytes: L };
    if (c.n) { o.n = hex(c.n, c.n.byteLength()); o.nbits = c.n.bitLength(); }
    if (c.type === 'short') {
      o.a = hex(c.a.fromRed(), L); o.b = hex(c.b.fromRed(), L);
      o.gx = hex(c.g.getX(), L); o.gy = hex(c.g.getY(), L);
      if (c.endo) {
        o.beta = hex(c.endo.beta.fromRed(), L);
        o.lambda = hex(c.endo.lambda, 32);
        o.basis = c.endo.basis.map(function(v) {
          return { a: v.a.toString(16), b: v.b.toString(16) };
        });
      }
    } else if (c.type === 'edwards') {
      o.a = hex(c.a.fromRed(), L); o.c = hex(c.c.fromRed(), L);
      o.d = hex(c.d.fromRed(), L);
      o.gx = hex(c.g.getX(), L); o.gy = hex(c.g.getY(), L);
    } else {
      o.a = hex(c.a.fromRed(), L); o.b = hex(c.b.fromRed(), L);
      o.gx = hex(c.g.getX(), L);
    }
    out[name] = o;
  });
  return out;
}

// ------------------------------------------------------------ mul_<curve>
function edgeScalars(c) {
  var n = c.n;
  var bits = c.type === 'short' ? flen(c) * 8 : 256;
  var list = [new BN(0), new BN(1), new BN(2), new BN(3), n.subn(1), n.clone(),
    n.addn(1), n.subn(2), n.ushrn(1), n.ushrn(1).addn(1),
    new BN(1).ushln(128), new BN(1).ushln(128).subn(1),
    new BN(1).ushln(bits - 1)];
  if (c.type !== 'short' || c.p.bitLength() % 8 === 0)
    list.push(new BN(1).ushln(bits).subn(1));       // all-ones, > n
  else
    list.push(new BN(1).ushln(c.p.bitLength()).subn(1));
  if (c.endo) list.push(c.endo.lambda.clone(), c.endo.lambda.addn(1),
    n.sub(c.endo.lambda));
  return list;
}

function genShortMul(name) {
  var pc = elliptic.curves[name];
  var c = pc.curve;
  var L = flen(c);
  var KL = L;                                   // scalar width at the C ABI
  var rng = new Prng('ellgpu-golden-v1:mul:' + name);
  var cases = [];
  var N = COUNTS[name];
  var i;
  var G = c.g;

  function rec(op, o) { o.op = op; cases.push(o); }
  function randPoint() { return G.mul(rng.below(c.n.subn(1)).addn(1)); }
  // a fresh (table-less) copy of G: forces the reference's variable-base path
  var G2 = c.point(G.getX(), G.getY());

  // fixed base: precomputed-G path (_fixedNafMul, base.js:52-84)
  edgeScalars(c).forEach(function(k) {
    rec('fixed', { k: hex(k, KL), r: affine(c, G.mul(k)) });
  });
  for (i = 0; i < N; i++) {
    var k = rng.below(c.n);
    rec('fixed', { k: hex(k, KL), r: affine(c, G.mul(k)) });
  }
  // variable base: _endoWnafMulAdd (short.js:218-249) / _wnafMul (base.js:86)
  var P0 = randPoint();
  edgeScalars(c).forEach(function(k) {
    rec('var', { k: hex(k, KL), px: hex(P0.getX(), L), py: hex(P0.getY(), L),
      r: affine(c, P0.mul(k)) });
  });
  edgeScalars(c).slice(0, 8).forEach(function(k) {
    rec('var', { k: hex(k, KL), px: hex(G2.getX(), L), py: hex(G2.getY(), L),
      r: affine(c, G2.mul(k)) });
  });
  for (i = 0; i < N; i++) {
    var P = randPoint();
    var kk = (i % 4 === 3) ? rng.bits(L * 8 > c.p.bitLength() ?
      c.p.bitLength() : L * 8) : rng.below(c.n);
    rec('var', { k: hex(kk, KL), px: hex(P.getX(), L), py: hex(P.getY(), L),
      r: affine(c, P.mul(kk)) });
  }
  // k1*P1 + k2*P2 (mulAdd, short.js:434-441); P1 = G exercises the mixed
  // precomputed / fresh path the verify uses.
  var one = new BN(1);
  var special = [
    [G, new BN(5), G2, new BN(7)],
    [P0, one, P0.neg(), one],                    // -> infinity
    [P0, one, P0, one],                          // -> doubling inside add
    [G, new BN(0), P0, new BN(0)],
    [G, new BN(0), P0, new BN(9)],
    [G, new BN(9), P0, new BN(0)],
    [G, c.n.subn(1), G2, one],                   // -G + G
    [G, c.n.subn(2), G2, new BN(2)],
    [G, new BN(2), G2.neg(), new BN(2)],
  ];
  special.forEach(function(s) {
    rec('muladd', { k1: hex(s[1], KL), p1x: hex(s[0].getX(), L),
      p1y: hex(s[0].getY(), L), k2: hex(s[3], KL), p2x: hex(s[2].getX(), L),
      p2y: hex(s[2].getY(), L), r: affine(c, s[0].mulAdd(s[1], s[2], s[3])) });
  });
  for (i = 0; i < Math.ceil(N / 2); i++) {
    var A = (i & 1) ? randPoint() : G;
    var B = randPoint();
    var k1 = rng.below(c.n);
    var k2 = rng.below(c.n);
    rec('muladd', { k1: hex(k1, KL), p1x: hex(A.getX(), L),
      p1y: hex(A.getY(), L), k2: hex(k2, KL), p2x: hex(B.getX(), L),
      p2y: hex(B.getY(), L), r: affine(c, A.mulAdd(k1, B, k2)) });
  }
  return cases;
}

function genEdwardsMul() {
  var name = 'ed25519';
  var c = elliptic.curves[name].curve;
  // as EDDSA's constructor does (eddsa/index.js:19): gives G its tables
  c.g.precompute(c.n.bitLength() + 1);
  var L = 32;
  var rng = new Prng('ellgpu-golden-v1:mul:' + name);
  var cases = [];
  var G = c.g;
  var N = COUNTS[name];
  var i;
  function rec(op, o) { o.op = op; cases.push(o); }
  function xy(p) { var a = affine(c, p); return a; }
  edgeScalars(c).forEach(function(k) {
    rec('fixed', { k: hex(k, L), r: affine(c, G.mul(k)) });
  });
  for (i = 0; i < N; i++) {
    var k = (i % 4 === 3) ? rng.bits(256) : rng.below(c.n);
    rec('fixed', { k: hex(k, L), r: affine(c, G.mul(k)) });
  }
  // variable base, including points with a torsion component (pointFromY of
  // a random y lands anywhere in the full group of order 8n)
  var pts = [];
  for (i = 0; i < N; i++) {
    if (i % 3 === 2) {
      for (;;) {
        var y = rng.below(c.p);
        try { pts.push(c.pointFromY(y, (i & 1) === 1)); break; } catch (e) { }
      }
    } else {
      var g0 = G.mul(rng.below(c.n.subn(1)).addn(1));
      pts.push(c.point(g0.getX(), g0.getY()));
    }
  }
  var P0 = pts[0];
  edgeScalars(c).forEach(function(k) {
    var a = xy(P0);
    rec('var', { k: hex(k, L), px: a.x, py: a.y, r: affine(c, P0.mul(k)) });
  });
  for (i = 0; i < N; i++) {
    var kk = (i % 4 === 1) ? rng.bits(256) : rng.below(c.n);
    var a2 = xy(pts[i]);
    rec('var', { k: hex(kk, L), px: a2.x, py: a2.y,
      r: affine(c, pts[i].mul(kk)) });
  }
  for (i = 0; i < N / 2; i++) {
    // reference quirk: Edwards mulAdd only works when one operand carries
    // precomputed tables (base.js:175 calls toJ(), which Edwards points lack),
    // so the first operand is always G here.
    var A = G;
    var B = pts[(i * 7 + 3) % N];
    var k1 = rng.below(c.n);
    var k2 = rng.below(c.n);
    var aa = xy(A);
    var bb = xy(B);
    rec('muladd', { k1: hex(k1, L), p1x: aa.x, p1y: aa.y, k2: hex(k2, L),
      p2x: bb.x, p2y: bb.y, r: affine(c, A.mulAdd(k1, B, k2)) });
  }
  return cases;
}

function genMontMul() {
  var name = 'curve25519';
  var c = elliptic.curves[name].curve;
  var L = 32;
  var rng = new Prng('ellgpu-golden-v1:mul:' + name);
  var cases = [];
  var G = c.g;
  var i;
  function res(p) {
    // reference: inf.getX() throws; record isInfinity instead
    if (p.isInfinity()) return { inf: true };
    return { x: hex(p.getX(), L) };
  }
  var ks = [new BN(0), new BN(1), new BN(2), new BN(6),
    new BN(1).ushln(255).subn(1), new BN(1).ushln(256).subn(1),
    new BN(1).ushln(254)];
  ks.forEach(function(k) {
    cases.push({ op: 'ladder', k: hex(k, L), px: hex(G.getX(), L),
      r: res(G.mul(k)) });
  });
  for (i = 0; i < COUNTS[name]; i++) {
    var x = (i & 1) ? rng.below(c.p) : G.mul(rng.bits(255)).getX();
    var P = c.point(x, new BN(1));
    var k = rng.bits(i % 4 === 2 ? 256 : 255);
    cases.push({ op: 'ladder', k: hex(k, L), px: hex(x, L), r: res(P.mul(k)) });
  }
  return cases;
}

// --------------------------------------------------------- verify_<curve>
function genVerify(name) {
  var pc = elliptic.curves[name];
  var c = pc.curve;
  var ec = new elliptic.ec(pc);
  var L = flen(c);
  var NL = c.n.byteLength();
  var rng = new Prng('ellgpu-golden-v1:verify:' + name);
  var N = Math.ceil(COUNTS[name] * 0.75);
  var cases = [];
  function rec(z, zlen, r, s, pub, note) {
    var zhex = Buffer.from(z).toString('hex');
    var ok = ec.verify(zhex, { r: r, s: s }, pub);
    cases.push({ z: zhex, r: hex(r, NL), s: hex(s, NL),
      qx: hex(pub.getPublic().getX(), L), qy: hex(pub.getPublic().getY(), L),
      ok: ok, note: note });
  }
  for (var i = 0; i < N; i++) {
    var key = ec.keyFromPrivate(hex(rng.below(c.n.subn(1)).addn(1), NL), 'hex');
    // hash lengths: the curve's natural digest, and longer/shorter ones to
    // exercise _truncateToN (ec/index.js:81-108)
    var zlen = [32, 32, 48, 64, 20][i % 5];
    var z = rng.bytes(zlen);
    if (i % 11 === 10) z[0] = 0;                 // leading-zero hash
    var sig = ec.sign(z, key);
    var pub = ec.keyFromPublic(key.getPublic());
    rec(z, zlen, sig.r, sig.s, pub, 'valid');
    var kind = i % 8;
    if (kind === 0) { var z2 = Buffer.from(z); z2[zlen - 1] ^= 1;
      rec(z2, zlen, sig.r, sig.s, pub, 'bad-z'); }
    if (kind === 1) rec(z, zlen, sig.r.xor(new BN(1).ushln(i % 100)), sig.s,
      pub, 'bad-r');
    if (kind === 2) rec(z, zlen, sig.r, sig.s.xor(new BN(2)), pub, 'bad-s');
    if (kind === 3) rec(z, zlen, sig.r, sig.s,
      ec.keyFromPublic(key.getPublic().add(c.g)), 'bad-q');
    if (kind === 4) rec(z, zlen, sig.r, c.n.sub(sig.s), pub, 'neg-s (valid)');
    if (kind === 5) rec(z, zlen, new BN(0), sig.s, pub, 'r=0');
    if (kind === 6) rec(z, zlen, sig.r, c.n.clone(), pub, 's=n');
    if (kind === 7) rec(z, zlen, c.n.clone(), new BN(0), pub, 'r=n,s=0');
  }
  return cases;
}

// ---------------------------------------------------- decompress_<curve>.json
// ShortCurve#pointFromX (short.js:187-204) / EdwardsCurve#pointFromY
// (edwards.js:71-97): valid and invalid abscissae, both parities.
function genDecompress(name) {
  var c = elliptic.curves[name].curve;
  var L = flen(c);
  var rng = new Prng('ellgpu-golden-v1:decompress:' + name);
  var cases = [];
  var N = COUNTS[name];
  function one(v, odd) {
    var o = { v: hex(v, L), odd: odd };
    try {
      var p = c.type === 'short' ? c.pointFromX(v, odd) : c.pointFromY(v, odd);
      o.r = affine(c, p);
      if (c.type !== 'short' && p.isInfinity()) o.r = { x: hex(p.getX(), L), y: hex(p.getY(), L) };
    } catch (e) {
      o.r = { invalid: e.message };
    }
    cases.push(o);
  }
  for (var i = 0; i < N; i++) {
    var P = c.g.mul(rng.below(c.n.subn(1)).addn(1));
    var v = c.type === 'short' ? P.getX() : P.getY();
    one(v, (i & 1) === 1);
    one(v, (i & 1) === 0);
    one(rng.below(c.p), (i & 2) === 2);            // ~half of these are not on the curve
  }
  [new BN(0), new BN(1), new BN(2), new BN(3), c.p.subn(1), c.p.subn(2), c.p.clone(), c.p.addn(1)]
    .forEach(function(v) {
      if (v.byteLength() > L) return;
      one(v, false); one(v, true);
    });
  return cases;
}

// EdwardsCurve#pointFromX (edwards.js:50-69): x -> (x, y), y of the requested parity
function genEdFromX() {
  var c = elliptic.curves.ed25519.curve;
  var rng = new Prng('ellgpu-golden-v1:fromx:ed25519');
  var cases = [];
  function one(v, odd) {
    var o = { v: hex(v, 32), odd: odd };
    try {
      var p = c.pointFromX(v, odd);
      o.r = { x: hex(p.getX(), 32), y: hex(p.getY(), 32) };
    } catch (e) { o.r = { invalid: e.message }; }
    cases.push(o);
  }
  for (var i = 0; i < 40; i++) {
    var P = c.g.mul(rng.below(c.n.subn(1)).addn(1));
    one(P.getX(), (i & 1) === 1);
    one(P.getX(), (i & 1) === 0);
    one(rng.below(c.p), (i & 2) === 2);            // ~half of these are no abscissa of the curve
  }
  [new BN(0), new BN(1), new BN(2), c.p.subn(1), c.p.subn(2), c.p.clone(), c.p.addn(1)]
    .forEach(function(v) { one(v, false); one(v, true); });
  return cases;
}

// ------------------------------------------------------------ codec_<curve>.json
// BaseCurve#decodePoint / BasePoint#encode (base.js:270-311), KeyPair#validate
// (ec/key.js:41-52); for ed25519 EDDSA#decodePoint / encodePoint (eddsa/index.js:94-109),
// EdwardsCurve#validate (edwards.js:99-112) and P.mul(n).isInfinity().
function genCodec(name) {
  var c = elliptic.curves[name].curve;
  var L = flen(c);
  var rng = new Prng('ellgpu-golden-v1:codec:' + name);
  var N = Math.max(8, COUNTS[name] >> 1);
  var out = { decode: [], encode: [], validate: [] };
  function tohex(arr) { return Buffer.from(arr).toString('hex'); }
  if (c.type === 'short') {
    var ec = new elliptic.ec(name);
    var dec = function(bytes) {
      var o = { enc: tohex(bytes) };
      try { o.r = affine(c, c.decodePoint(bytes)); } catch (e) { o.r = { throws: e.message }; }
      out.decode.push(o);
    };
    var val = function(x, y) {
      var o = { x: hex(x, L), y: hex(y, L) };
      var v = ec.keyFromPublic({ x: o.x, y: o.y }).validate();
      o.result = v.result; o.reason = v.reason;
      out.validate.push(o);
    };
    for (var i = 0; i < N; i++) {
      var P = c.g.mul(rng.below(c.n.subn(1)).addn(1));
      var un = P.encode('array', false), co = P.encode('array', true);
      out.encode.push({ x: hex(P.getX(), L), y: hex(P.getY(), L), compact: tohex(co), full: tohex(un) });
      dec(un);
      var odd = P.getY().isOdd();
      dec([odd ? 7 : 6].concat(un.slice(1)));             // hybrid, consistent
      dec([odd ? 6 : 7].concat(un.slice(1)));             // hybrid, contradicting -> assert
      dec(co); dec([co[0] ^ 1].concat(co.slice(1)));
      dec([[0, 1, 5, 8, 0xff][i % 5]].concat(un.slice(1)));     // unknown prefix, long form
      dec([[0, 4, 6, 0x12][i % 4]].concat(co.slice(1)));       // long-form prefix on a short string
      dec([[2, 3][i % 2]].concat(un.slice(1)));                 // short-form prefix on a long string
      // not on the curve: decodePoint does not check, validate does
      var bx = rng.below(c.p), by = rng.below(c.p);
      dec([4].concat(bx.toArray('be', L), by.toArray('be', L)));
      val(P.getX(), P.getY());
      val(bx, by);
      val(P.getX(), P.getY().addn(1).umod(c.p));
      dec([2 + (i & 1)].concat(rng.below(c.p).toArray('be', L)));   // ~half invalid
    }
    // coordinates >= p are reduced by Point's toRed
    if (c.p.bitLength() % 8 === 0 || name === 'p521') {
      var big = new BN(1).ushln(L * 8).subn(1 + 5);
      dec([4].concat(big.toArray('be', L), c.p.addn(3).toArray('be', L)));
      dec([2].concat(c.p.addn(1).toArray('be', L))); dec([3].concat(big.toArray('be', L)));
      val(c.g.getX().add(c.p), c.g.getY());
    }
    val(new BN(0), new BN(0));
    val(c.g.getX(), c.g.getY());
    val(c.g.getX(), c.p.sub(c.g.getY()));
  } else {
    var ed = new elliptic.eddsa(name);
    var edec = function(bytes) {
      var o = { enc: tohex(bytes) };
      try {
        var p = ed.decodePoint(bytes);
        o.r = { x: hex(p.getX(), L), y: hex(p.getY(), L) };
      } catch (e) { o.r = { throws: e.message }; }
      out.decode.push(o);
    };
    var eval_ = function(p) {
      var o = { x: hex(p.getX(), L), y: hex(p.getY(), L) };
      o.on_curve = c.validate(c.point(o.x, o.y));
      o.order_ok = c.point(o.x, o.y).mul(c.n).isInfinity();
      out.validate.push(o);
    };
    // a point of order 8 (y from the well-known small-order encodings), via decodePoint
    var t8 = ed.decodePoint('26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05');
    for (var j = 0; j < N; j++) {
      var Q = c.g.mul(rng.below(c.n.subn(1)).addn(1));
      var enc = ed.encodePoint(Q);
      out.encode.push({ x: hex(Q.getX(), L), y: hex(Q.getY(), L), compact: tohex(enc) });
      edec(enc);
      edec(enc.slice(0, 31).concat(enc[31] ^ 0x80));
      edec(Array.prototype.slice.call(rng.bytes(32)));                // ~half invalid
      eval_(Q);
      var M = Q.add(j & 1 ? t8 : t8.dbl());                          // mixed order: on curve, n*M != O
      eval_(M);
      out.validate.push({ x: hex(Q.getX(), L), y: hex(Q.getY().addn(1).umod(c.p), L), on_curve: false, order_ok: null });
    }
    eval_(t8); eval_(t8.dbl()); eval_(t8.dbl().dbl());
    eval_(c.point(null, null, null));                                 // identity (0, 1)
    ['0100000000000000000000000000000000000000000000000000000000000000',
      '0100000000000000000000000000000000000000000000000000000000000080',
      '0000000000000000000000000000000000000000000000000000000000000000',
      'ecffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f',
      'edffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff7f',
      'eeffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff',
      'ffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffffff'].forEach(function(h) {
      edec(Array.prototype.slice.call(Buffer.from(h, 'hex')));
    });
    // the third off-curve entry above was not produced by the reference's validate: fix that
    out.validate.forEach(function(o) {
      if (o.order_ok === null) { o.on_curve = c.validate(c.point(o.x, o.y)); delete o.order_ok; }
    });
  }
  return out;
}

// ------------------------------------------------------------- wire_<curve>.json
// EC#verify(msg, derSignature, encodedKey, 'hex') (ec/index.js:188-229 through keyFromPublic /
// decodePoint and new Signature(der)), Signature#toDER / _importDER (ec/signature.js).
//   verify: { z, der, key, ok } or { ..., throws: message }
//   der:    { r, s, der } from toDER;   parse: { der, r, s } or { der, bad: true }
function genWire(name) {
  var pc = elliptic.curves[name];
  var c = pc.curve;
  var ec = new elliptic.ec(pc);
  var NL = c.n.byteLength();
  var rng = new Prng('ellgpu-golden-v1:wire:' + name);
  var N = Math.max(12, COUNTS[name] >> 1);
  var out = { verify: [], der: [], parse: [] };
  var Signature = ec.sign(rng.bytes(32), ec.keyFromPrivate('01', 'hex')).constructor;
  function ver(z, der, key, note) {
    var o = { z: Buffer.from(z).toString('hex'), der: Buffer.from(der).toString('hex'),
      key: Buffer.from(key).toString('hex'), note: note };
    try { o.ok = ec.verify(o.z, o.der, o.key, 'hex'); } catch (e) { o.throws = e.message; }
    out.verify.push(o);
  }
  function parse(der) {
    var o = { der: Buffer.from(der).toString('hex') };
    var sg = Object.create(Signature.prototype);
    if (sg._importDER(o.der, 'hex')) { o.r = sg.r.toString(16); o.s = sg.s.toString(16); } else o.bad = true;
    out.parse.push(o);
  }
  function toDer(r, s) {
    var d = new Signature({ r: r, s: s }).toDER();
    out.der.push({ r: hex(r, NL), s: hex(s, NL), der: Buffer.from(d).toString('hex') });
    return d;
  }
  function intDer(bytes) { return [2, bytes.length].concat(bytes); }
  function seq(body) {
    return body.length < 128 ? [0x30, body.length].concat(body) : [0x30, 0x81, body.length].concat(body);
  }
  for (var i = 0; i < N; i++) {
    var key = ec.keyFromPrivate(hex(rng.below(c.n.subn(1)).addn(1), NL), 'hex');
    var z = rng.bytes(32);
    var sig = ec.sign(z, key, { canonical: (i & 1) === 1 });
    var der = toDer(sig.r, sig.s);
    var compressed = (i % 3) !== 0;
    var pk = key.getPublic().encode('array', compressed);
    ver(z, der, pk, 'valid');
    parse(der);
    var k = i % 12;
    var bad;
    if (k === 0) { bad = der.slice(); bad[0] = 0x31; ver(z, bad, pk, 'bad-seq-tag'); parse(bad); }
    if (k === 1) { bad = der.concat([0]); ver(z, bad, pk, 'trailing-byte'); parse(bad); }
    if (k === 2) { bad = der.slice(0, der.length - 1); ver(z, bad, pk, 'truncated'); parse(bad); }
    if (k === 3) { bad = der.slice(); bad[der[1] & 0x80 ? 2 : 1] ^= 1; ver(z, bad, pk, 'bad-length'); parse(bad); }
    if (k === 4) {                                   // non-minimal integer: extra leading zero
      bad = seq(intDer([0].concat(sig.r.toArray('be', NL))).concat(intDer([0].concat(sig.s.toArray('be', NL)))));
      ver(z, bad, pk, 'zero-padded'); parse(bad);
    }
    if (k === 5) {                                   // negative integer (high bit without the pad)
      var rb = sig.r.toArray(); rb[0] |= 0x80;
      bad = seq(intDer(rb).concat(intDer(sig.s.toArray('be', NL + 1))));
      ver(z, bad, pk, 'negative-r'); parse(bad);
    }
    if (k === 6) {                                   // long-form length where the short form fits
      var body = intDer(sig.r.toArray('be', NL + 1)).concat(intDer(sig.s.toArray('be', NL + 1)));
      if (sig.r.toArray('be', NL)[0] & 0x80 && sig.s.toArray('be', NL)[0] & 0x80 && body.length < 128) {
        bad = [0x30, 0x81, body.length].concat(body); ver(z, bad, pk, 'long-form-len'); parse(bad);
      } else { bad = der.slice(); bad[der.length - 1] ^= 4; ver(z, bad, pk, 'bad-s'); parse(bad); }
    }
    if (k === 7) {                                   // r wider than n: well-formed, verify false
      var wide = [1].concat(sig.r.toArray('be', NL));
      bad = seq(intDer(wide).concat(intDer(sig.s.toArray())));
      if (sig.s.toArray()[0] & 0x80) bad = seq(intDer(wide).concat(intDer([0].concat(sig.s.toArray()))));
      ver(z, bad, pk, 'wide-r'); parse(bad);
    }
    if (k === 8) { var z2 = Buffer.from(z); z2[31] ^= 1; ver(z2, der, pk, 'bad-z'); }
    if (k === 9) {                                   // key errors come first
      var bk = pk.slice(); bk[0] = 5; ver(z, der, bk, 'bad-key-prefix');
      bad = der.slice(); bad[0] = 0x31; ver(z, bad, bk, 'bad-key-and-sig');
    }
    if (k === 10) {
      var un = key.getPublic().encode('array', false);
      var odd = key.getPublic().getY().isOdd();
      ver(z, der, [odd ? 7 : 6].concat(un.slice(1)), 'hybrid-key');
      ver(z, der, [odd ? 6 : 7].concat(un.slice(1)), 'hybrid-key-mismatch');
      ver(z, der, [2].concat(rng.below(c.p).toArray('be', c.p.byteLength())), 'random-x-key');
    }
    if (k === 11) {
      ver(z, seq(intDer([0]).concat(intDer(sig.s.toArray('be', NL + 1).slice(sig.s.toArray('be', NL)[0] & 0x80 ? 0 : 1)))), pk, 'r=0 one byte');
      ver(z, seq([2, 0].concat(intDer([1]))), pk, 'empty r');
      ver(z, [], pk, 'empty');
      ver(z, [0x30], pk, 'only tag');
      ver(z, [0x30, 0x80], pk, 'indefinite');
      ver(z, [0x30, 0x85, 1, 0, 0, 0, 6, 2, 1, 1, 2, 1, 1], pk, 'five length octets');
      ver(z, [0x30, 0x06, 2, 1, 1, 2, 1, 1], pk, 'tiny r,s');
      ver(z, [0x30, 0x82, 0, 6, 2, 1, 1, 2, 1, 1], pk, 'length with zero octet');
      parse([0x30, 0x06, 2, 1, 1, 2, 1, 1]); parse([0x30, 0x82, 0, 6, 2, 1, 1, 2, 1, 1]);
      parse([0x30, 0x81, 6, 2, 1, 1, 2, 1, 1]); parse([0x30, 0x06, 2, 0x81, 1, 1, 2, 1, 1]);
      parse([]); parse([0x30]); parse([0x30, 4, 2, 0, 2, 0]); parse([0x30, 5, 2, 1, 0, 2, 0]);
      parse([0x30, 6, 2, 1, 0x7f, 2, 1, 0]); parse([0x30, 7, 2, 2, 0, 0x80, 2, 1, 5]);
    }
    // toDER on small / edge values
    toDer(new BN(i + 1), new BN(1).ushln(8 * (i % NL)).addn(i));
    toDer(new BN(1).ushln(8 * NL - 1 - (i % 9)), rng.below(c.n.subn(1)).addn(1));
  }
  toDer(new BN(0), new BN(1));
  toDer(c.n.subn(1), c.n.subn(1));
  toDer(new BN(0x80), new BN(0x7f));
  return out;
}

// -------------------------------------------------------------- add_<curve>.json
// Point#add (short.js:365-412 / edwards.js:350-360): random pairs, P + P, P + (-P), O on
// either side, points of order 2 where the curve has them, and (short curves) off-curve
// operands -- the reference's chord / tangent formulas define those results too.
function genAdd(name) {
  var c = elliptic.curves[name].curve;
  var L = flen(c);
  var rng = new Prng('ellgpu-golden-v1:add:' + name);
  var N = Math.max(10, COUNTS[name] >> 1);
  var cases = [];
  function enc(p) {
    if (c.type === 'short') return p.inf ? { inf: true } : { x: hex(p.getX(), L), y: hex(p.getY(), L) };
    return { x: hex(p.getX(), L), y: hex(p.getY(), L) };
  }
  function one(p, q, note) {
    var r = p.add(q);
    var o = { p: enc(p), q: enc(q), note: note };
    if (c.type === 'short') o.r = r.inf ? { inf: true } : { x: hex(r.getX(), L), y: hex(r.getY(), L) };
    else o.r = { x: hex(r.getX(), L), y: hex(r.getY(), L), inf: r.isInfinity() };
    cases.push(o);
  }
  var O = c.type === 'short' ? c.point(null, null) : c.point(null, null, null);
  for (var i = 0; i < N; i++) {
    var P = c.g.mul(rng.below(c.n.subn(1)).addn(1));
    var Q = c.g.mul(rng.below(c.n.subn(1)).addn(1));
    if (c.type !== 'short') { P = c.point(P.getX(), P.getY()); Q = c.point(Q.getX(), Q.getY()); }
    one(P, Q, 'random');
    if (i % 4 === 0) one(P, P, 'P+P');
    if (i % 4 === 1) one(P, P.neg(), 'P+(-P)');
    if (i % 4 === 2) { one(P, O, 'P+O'); one(O, Q, 'O+Q'); }
    if (i % 4 === 3) one(O, O, 'O+O');
    if (c.type === 'short') {
      var bx = rng.below(c.p), by = rng.below(c.p), cx = rng.below(c.p), cy = rng.below(c.p);
      one(c.point(bx, by), c.point(cx, cy), 'off-curve');
      if (i % 3 === 0) one(c.point(bx, by), c.point(bx, by), 'off-curve P+P');
      if (i % 3 === 1) one(c.point(bx, by), c.point(bx, cy), 'off-curve same x');
      if (i % 3 === 2) one(c.point(bx, new BN(0)), c.point(bx, new BN(0)), 'y = 0 doubled');
    }
  }
  if (c.type !== 'short') {
    var t8 = new elliptic.eddsa(name).decodePoint('26e8958fc2b227b045c3f489f2ef98f0d5dfac05d3c63339b13802886d53fc05');
    t8 = c.point(t8.getX(), t8.getY());
    var t4 = t8.dbl(); t4 = c.point(t4.getX(), t4.getY());
    var t2 = t4.dbl(); t2 = c.point(t2.getX(), t2.getY());
    one(t8, t8, 'order 8 doubled'); one(t4, t4, 'order 4 doubled'); one(t2, t2, 'order 2 doubled');
    one(t8, c.point(c.g.getX(), c.g.getY()), 'order 8 + G');
  }
  return cases;
}


// ---------------------------------------------------------- offcurve_<curve>.json
// Points that are NOT on the curve.  The reference never validates them on this path
// (ec/index.js:192 keyFromPublic, ec/key.js:27-35, short.js:422-432, edwards.js:362-367): it
// runs its formulas on whatever (x, y) it is given, and the result depends on the exact order
// of its operations (off the curve the "group law" is not associative).  These fixtures pin
// what it answers, so that (i) the engine's C ABI can be shown to report such items as
// outside its domain instead of guessing, and (ii) the patched library can be shown to return
// the reference's own result for them.
//   var     k * P, P off the curve (incl. y = 0, k = 0, 1, n)
//   muladd  k1 * P1 + k2 * P2 with one or both points off the curve (g1: P1 is the curve's G)
//   verify  ECDSA tuples over an off-curve key built so that the reference ANSWERS TRUE
//           (R = u1 G + u2 Q as the reference computes it, r = R.x mod n, s = r / u2,
//           z = u1 s), plus corrupted ones it answers false, plus r / s out of range
//   on      control items (on-curve points) mixed in: the engine must still compute those
//   add     (ed25519) Point#add of off-curve operands: one formula, equal in the engine
function genOffCurve(name) {
  var pc = elliptic.curves[name];
  var c = pc.curve;
  var short = c.type === 'short';
  if (!short) c.g.precompute(c.n.bitLength() + 1);
  var L = flen(c);
  var rng = new Prng('ellgpu-golden-v1:offcurve:' + name);
  var N = Math.max(8, COUNTS[name] >> 2);
  var cases = [];
  var G = c.g;
  function rec(op, o) { o.op = op; cases.push(o); }
  function offPoint() {
    for (;;) {
      var x = rng.below(c.p), y = rng.below(c.p);
      var P = c.point(x, y);
      if (!c.validate(P)) return P;
    }
  }
  function onPoint() {
    var P = G.mul(rng.below(c.n.subn(1)).addn(1));
    return c.point(P.getX(), P.getY());
  }
  function xy(P) { return affine(c, P); }
  var i, k, P, a;
  // k * P
  var P0 = offPoint();
  [new BN(0), new BN(1), new BN(2), c.n.clone(), c.n.subn(1)].forEach(function(k) {
    a = xy(P0);
    rec('var', { k: hex(k, L), px: a.x, py: a.y, on: false, r: affine(c, P0.mul(k)) });
  });
  if (short) {
    // a point with y = 0 (order 2 on "its" curve)
    var Z = c.point(rng.below(c.p), new BN(0));
    [new BN(1), new BN(2), new BN(3), rng.below(c.n)].forEach(function(k) {
      rec('var', { k: hex(k, L), px: hex(Z.getX(), L), py: hex(Z.getY(), L), on: false,
        r: affine(c, Z.mul(k)) });
    });
  }
  for (i = 0; i < N; i++) {
    var on = i % 4 === 3;
    P = on ? onPoint() : offPoint();
    k = rng.below(c.n);
    a = xy(P);
    rec('var', { k: hex(k, L), px: a.x, py: a.y, on: on, r: affine(c, P.mul(k)) });
  }
  // k1 * P1 + k2 * P2.  Edwards mulAdd works in the reference only with a precomputed first
  // operand (base.js:175 calls toJ()), so P1 = G there.
  for (i = 0; i < N; i++) {
    var kind = short ? i % 5 : 0;        // 0: G + off, 1: on + off, 2: off + on, 3: off + off, 4: on + on (control)
    var A = kind === 0 ? G : ((kind === 1 || kind === 4) ? onPoint() : offPoint());
    var B = (kind === 2 || kind === 4) ? onPoint() : offPoint();
    var k1 = rng.below(c.n), k2 = rng.below(c.n);
    var aa = xy(A), bb = xy(B);
    rec('muladd', { k1: hex(k1, L), p1x: aa.x, p1y: aa.y, k2: hex(k2, L), p2x: bb.x, p2y: bb.y,
      g1: kind === 0, on: kind === 4, r: affine(c, A.mulAdd(k1, B, k2)) });
  }
  if (!short) {
    // Point#add (edwards.js:350-360 -> _extAdd): ONE formula, so the engine's ellgpu_point_add
    // has to give the reference's coordinates off the curve as well
    for (i = 0; i < N; i++) {
      var U = offPoint(), V = (i % 3 === 0) ? U : (i % 3 === 1 ? onPoint() : offPoint());
      var S = U.add(V);
      var ua = xy(U), va = xy(V);
      rec('add', { p: { x: ua.x, y: ua.y }, q: { x: va.x, y: va.y },
        r: { x: hex(S.getX(), L), y: hex(S.getY(), L) } });
    }
    return cases;
  }
  // ECDSA over off-curve keys
  var ec = new elliptic.ec(pc);
  var NL = c.n.byteLength();
  var ZL = (c.n.bitLength() % 8 === 0) ? NL : NL - 1;      // digest bytes that _truncateToN leaves alone
  function verifyCase(Q, z, r, s, note) {
    var zhex = z.toString(16, ZL * 2);
    var key = ec.keyFromPublic({ x: hex(Q.getX(), L), y: hex(Q.getY(), L) });
    var ok = ec.verify(zhex, { r: r, s: s }, key);
    rec('verify', { z: zhex, r: hex(r, NL), s: hex(s, NL), qx: hex(Q.getX(), L), qy: hex(Q.getY(), L),
      ok: ok, on: c.validate(Q), note: note });
    return ok;
  }
  var made = 0;
  for (i = 0; made < N && i < 20 * N; i++) {
    var Q = offPoint();
    var u1 = rng.below(c.n.subn(1)).addn(1), u2 = rng.below(c.n.subn(1)).addn(1);
    var R = G.mulAdd(u1, Q, u2);
    if (R.isInfinity()) continue;
    var r = R.getX().umod(c.n);
    if (r.isZero()) continue;
    var s = r.mul(u2.invm(c.n)).umod(c.n);
    var z = u1.mul(s).umod(c.n);
    if (s.isZero() || z.byteLength() > ZL) continue;
    if (!verifyCase(Q, z, r, s, 'off-curve key, reference says true'))
      throw new Error('construction failed: the reference rejected its own R');
    made++;
    var m = made % 4;
    if (m === 0) verifyCase(Q, z.xor(new BN(1)), r, s, 'off-curve key, bad z');
    if (m === 1) verifyCase(Q, z, new BN(0), s, 'off-curve key, r = 0');
    if (m === 2) verifyCase(Q, z, r, c.n.clone(), 'off-curve key, s = n');
    if (m === 3) {
      // control: a real signature under an on-curve key
      var key = ec.keyFromPrivate(hex(rng.below(c.n.subn(1)).addn(1), NL), 'hex');
      var zz = new BN(rng.bytes(ZL));
      if (c.n.bitLength() % 8 !== 0) zz = zz.maskn(8 * ZL);
      var sig = ec.sign(zz.toString(16, ZL * 2), key);
      verifyCase(key.getPublic(), zz, sig.r, sig.s, 'on-curve control');
    }
  }
  if (made < N) throw new Error('too few off-curve verify tuples for ' + name);
  return cases;
}

// ------------------------------------------------------- der_fuzz_secp256k1.json
// Signature#_importDER on mutated encodings: byte flips, insertions, deletions, length-field
// edits and splices of valid signatures (seeded).  { der, r, s } or { der, bad: true }.
function genDerFuzz() {
  var ec = new elliptic.ec('secp256k1');
  var c = ec.curve;
  var rng = new Prng('ellgpu-golden-v1:derfuzz');
  var Signature = ec.sign(rng.bytes(32), ec.keyFromPrivate('01', 'hex')).constructor;
  var out = [];
  var seen = {};
  function rec(bytes) {
    var h = Buffer.from(bytes).toString('hex');
    if (seen[h]) return;
    seen[h] = true;
    var o = { der: h };
    var sg = Object.create(Signature.prototype);
    if (sg._importDER(h, 'hex')) { o.r = sg.r.toString(16); o.s = sg.s.toString(16); } else o.bad = true;
    out.push(o);
  }
  function rnd(n) { return rng.bytes(2).readUInt16BE(0) % n; }
  for (var i = 0; i < 420; i++) {
    // r, s of assorted widths so that pads, short integers and long forms all occur
    var rb = 1 + rnd(40), sb = 1 + rnd(40);
    var r = new BN(rng.bytes(rb)), s2 = new BN(rng.bytes(sb)).addn(1);
    var der = new Signature({ r: r, s: s2 }).toDER();
    rec(der);
    for (var m = 0; m < 5; m++) {
      var d = der.slice();
      var kind = rnd(7);
      var pos = rnd(d.length);
      if (kind === 0) d[pos] ^= 1 << rnd(8);
      else if (kind === 1) d.splice(pos, 1);
      else if (kind === 2) d.splice(pos, 0, rng.bytes(1)[0]);
      else if (kind === 3) d[Math.min(pos, 5)] = [0, 0x80, 0x81, 0x82, 0x84, 0x85, 0xff, 0x7f][rnd(8)];
      else if (kind === 4) d = d.slice(0, pos);
      else if (kind === 5) d = d.concat(Array.prototype.slice.call(rng.bytes(1 + rnd(3))));
      else { d[1] = 0x81; d.splice(2, 0, d.length - 2); }       // long-form outer length
      rec(d);
    }
  }
  return out;
}

// ------------------------------------------------------ eddsa_verify_ed25519.json
// EDDSA#verify (eddsa/index.js:52-63) on the official ed25519 sign.input vectors the
// reference ships (test/fixtures/sign.input), plus corrupted / malformed variants.
// Result: ok = true|false, or throws = message when the reference throws.
function genEddsa() {
  var ed = new elliptic.eddsa('ed25519');
  var rng = new Prng('ellgpu-golden-v1:eddsa');
  var lines = fs.readFileSync(path.join(ref.root, 'test', 'fixtures', 'sign.input'), 'utf8')
    .split('\n').filter(function(l) { return l.length; });
  var cases = [];
  function rec(msg, sig, pub, note) {
    var o = { msg: msg, sig: sig, pub: pub, note: note };
    try { o.ok = ed.verify(msg.length ? Buffer.from(msg, 'hex').toJSON().data : [], sig, pub); }
    catch (e) { o.throws = e.message; }
    cases.push(o);
  }
  function flip(hexs, byte, bit) {
    var b = Buffer.from(hexs, 'hex'); b[byte] ^= 1 << bit; return b.toString('hex');
  }
  var picks = [];
  for (var i = 0; i < 96; i++) picks.push(i);
  [127, 128, 129, 200, 255, 256, 400, 511, 512, 700, 1023].forEach(function(i) { picks.push(i); });
  picks.forEach(function(i, j) {
    var f = lines[i].split(':');
    var pub = f[1];
    var msg = f[2];
    var sig = f[3].slice(0, 128);
    rec(msg, sig, pub, 'sign.input line ' + i);
    var kind = j % 8;
    if (kind === 0) rec(msg.length ? flip(msg, 0, 0) : '00', sig, pub, 'bad msg');
    if (kind === 1) rec(msg, flip(sig, 40, 3), pub, 'bad S');
    if (kind === 2) rec(msg, flip(sig, 5, 1), pub, 'bad R');
    if (kind === 3) rec(msg, sig, flip(pub, 7, 2), 'bad A');
    if (kind === 4) {                                  // S + n (>= n): must be rejected
      var S = new BN(Buffer.from(sig.slice(64), 'hex'), 'le').add(ed.curve.n);
      if (S.byteLength() <= 32)
        rec(msg, sig.slice(0, 64) + Buffer.from(S.toArray('le', 32)).toString('hex'), pub, 'S+n');
    }
    if (kind === 5) rec(msg, flip(sig, 31, 7), pub, 'R sign bit flipped');
    if (kind === 6) rec(msg, sig, flip(pub, 31, 7), 'A sign bit flipped');
    if (kind === 7) rec(msg, rng.bytes(64).toString('hex'), pub, 'random sig');
  });
  // Small-order / non-canonical R: decodePoint reduces y mod p, so y = p + 1 (the identity,
  // non-canonically) and y = p (the order-4 point (sqrt(-1), 0)... y = 0) are accepted as
  // encodings.  Signatures with such an R are built by hand: S = h * a (R = identity: r = 0).
  var p = ed.curve.p;
  for (var t = 0; t < 6; t++) {
    var key = ed.keyFromSecret(rng.bytes(32).toString('hex'));
    var a = key.priv();
    var m = rng.bytes(10 + t);
    [p.addn(1), new BN(1)].forEach(function(y, which) {          // identity: non-canonical, canonical
      var renc = y.toArray('le', 32);
      var h = ed.hashInt(renc, key.pubBytes(), Array.prototype.slice.call(m));
      var S = h.mul(a).umod(ed.curve.n);
      var sg = Buffer.from(renc.concat(S.toArray('le', 32))).toString('hex');
      rec(m.toString('hex'), sg, Buffer.from(key.pubBytes()).toString('hex'),
        which === 0 ? 'R = identity, non-canonical y = p + 1' : 'R = identity, canonical');
      if (t === 0) rec(m.toString('hex'), flip(sg, 40, 1), Buffer.from(key.pubBytes()).toString('hex'), 'same, bad S');
    });
    // non-canonical y in [p, 2^255): y = p + k for small k; valid only by accident, must agree anyway
    var renc2 = p.addn(3 + t).toArray('le', 32);
    rec(m.toString('hex'), Buffer.from(renc2.concat(rng.bytes(32).toJSON().data)).toString('hex').slice(0, 126) + '00',
      Buffer.from(key.pubBytes()).toString('hex'), 'R non-canonical y = p + ' + (3 + t));
  }
  return cases;
}

// ------------------------------------------------------------ signdet_<curve>.json
// EC#sign (ec/index.js:110-186) with its own nonces: HmacDRBG (hmac-drbg 1.0.1) over the
// curve's hash, entropy = the private key, nonce = the truncated message, both n.byteLength()
// bytes -- i.e. deterministic signatures.  (z, d, canonical) -> (r, s, recoveryParam).
function genSignDet(name) {
  var pc = elliptic.curves[name];
  var ec = new elliptic.ec(pc);
  var c = pc.curve;
  var NB = c.n.byteLength();
  var rng = new Prng('ellgpu-golden-v1:signdet:' + name);
  var cases = [];
  function one(z, d, canonical, note) {
    var sig = ec.sign(z, ec.keyFromPrivate(hex(d, NB), 'hex'), { canonical: canonical });
    cases.push({ z: Buffer.from(z).toString('hex'), d: hex(d, NB), canonical: canonical,
      r: hex(sig.r, NB), s: hex(sig.s, NB), recid: sig.recoveryParam, note: note });
  }
  var N = Math.ceil(COUNTS[name] / 2);
  for (var i = 0; i < N; i++) {
    var zlen = [32, NB, 20, 48, 64, 32][i % 6];
    one(rng.bytes(zlen), rng.below(c.n.subn(1)).addn(1), (i & 1) === 1, 'seeded, ' + zlen + '-byte digest');
  }
  var d0 = rng.below(c.n.subn(1)).addn(1);
  one(Buffer.alloc(32), d0, true, 'digest 0');
  one(Buffer.alloc(NB, 0xff), d0, false, 'digest ff..ff');
  one(Buffer.from(c.n.toArray('be', NB)), d0, true, 'digest = n');
  one(Buffer.from(c.n.subn(1).toArray('be', NB)), d0, false, 'digest = n - 1');
  one(rng.bytes(32), new BN(1), true, 'd = 1');
  one(rng.bytes(32), c.n.subn(1), false, 'd = n - 1');
  return cases;
}

// ------------------------------------------------------------ recover_<curve>.json
// EC#recoverPubKey (ec/index.js:231-259): (e, r, s, j) -> Q = r^-1 (s R - e G) with R the point
// of x-coordinate r (+ n when j & 2) and y-parity j & 1.  Real signatures with all four j,
// random (r, s), small r (the second candidate exists only for r < p - n), digests longer
// than n (e is NOT truncated here, only reduced).  Result: point, inf, or the message thrown.
function genRecover(name) {
  var pc = elliptic.curves[name];
  var ec = new elliptic.ec(pc);
  var c = pc.curve;
  var NB = c.n.byteLength();
  var rng = new Prng('ellgpu-golden-v1:recover:' + name);
  var cases = [];
  function one(z, r, s, j, note) {
    var o = { z: Buffer.from(z).toString('hex'), r: hex(r, NB), s: hex(s, NB), j: j, note: note };
    try {
      var Q = ec.recoverPubKey(z, { r: hex(r, NB), s: hex(s, NB) }, j);
      o.q = affine(c, Q);
    } catch (e) {
      o.throws = e.message;
    }
    cases.push(o);
  }
  var N = Math.ceil(COUNTS[name] / 3);
  for (var i = 0; i < N; i++) {
    var zlen = [32, 32, NB, 20, 48][i % 5];
    if (zlen > 2 * NB) zlen = NB;
    var z = rng.bytes(zlen);
    var key = ec.keyFromPrivate(hex(rng.below(c.n.subn(1)).addn(1), NB), 'hex');
    var sig = ec.sign(z, key, { canonical: (i & 1) === 1 });
    for (var j = 0; j < 4; j++) one(z, sig.r, sig.s, j, j === sig.recoveryParam ? 'signature, its own j' : 'signature, other j');
    var last = cases[cases.length - 4 + sig.recoveryParam];
    // note: EC#sign truncates the digest (_truncateToN), recoverPubKey does not: they only agree
    // when the digest is not longer than n
    if (zlen * 8 <= c.n.bitLength() && (!last.q || last.q.x !== hex(key.getPublic().getX(), flen(c))))
      throw new Error('reference recoverPubKey disagrees with the signing key');
  }
  for (i = 0; i < N; i++) {
    one(rng.bytes(32), rng.below(c.n.subn(1)).addn(1), rng.below(c.n.subn(1)).addn(1), i & 3, 'random r, s');
  }
  // small r: the second candidate r + n is below p
  var pmn = c.p.umod(c.n);
  for (i = 0; i < 12; i++) {
    var rs = rng.below(pmn.subn(1)).addn(1);
    one(rng.bytes(32), rs, rng.below(c.n.subn(1)).addn(1), 2 + (i & 1), 'r < p mod n, second candidate');
  }
  one(rng.bytes(32), pmn, new BN(5), 2, 'r = p mod n, second candidate');
  one(rng.bytes(32), pmn.subn(1), new BN(5), 3, 'r = p mod n - 1, second candidate');
  one(Buffer.alloc(32), new BN(7), new BN(9), 0, 'e = 0');
  one(Buffer.from(c.n.toArray('be', NB)), new BN(7), new BN(9), 1, 'e = n');
  one(Buffer.from(c.n.addn(1).toArray('be', NB)), c.n.subn(1), c.n.subn(1), 1, 'e = n + 1, r = s = n - 1');
  one(rng.bytes(2 * NB > 64 ? 64 : 2 * NB), rng.below(c.n.subn(1)).addn(1), new BN(3), 0, 'long digest');
  one(rng.bytes(32), new BN(1), new BN(1), 0, 'r = s = 1');
  return cases;
}

// ------------------------------------------------------ eddsa_sign_ed25519.json
// EDDSA#sign (eddsa/index.js:32-50) with KeyPair.fromSecret (eddsa/key.js:42-75): a pure
// function (secret, message) -> (signature, public key).  The official sign.input vectors
// (secret, pub, msg, sig) plus seeded secrets with message lengths around the SHA-512 block
// boundaries of both hashes (prefix||M: 32 + len, R||A||M: 64 + len; padding flips at
// 112 mod 128).
function genEddsaSign() {
  var ed = new elliptic.eddsa('ed25519');
  var rng = new Prng('ellgpu-golden-v1:eddsa-sign');
  var lines = fs.readFileSync(path.join(ref.root, 'test', 'fixtures', 'sign.input'), 'utf8')
    .split('\n').filter(function(l) { return l.length; });
  var cases = [];
  function rec(secretHex, msgHex, note) {
    var key = ed.keyFromSecret(secretHex);
    var msg = msgHex.length ? Buffer.from(msgHex, 'hex').toJSON().data : [];
    var sig = key.sign(msg).toHex().toLowerCase();
    cases.push({ secret: secretHex, msg: msgHex, sig: sig,
      pub: Buffer.from(key.getPublic()).toString('hex'), note: note });
  }
  var picks = [];
  for (var i = 0; i < 64; i++) picks.push(i);
  [100, 127, 128, 129, 255, 256, 511, 512, 1000, 1023].forEach(function(i) { picks.push(i); });
  picks.forEach(function(i) {
    var f = lines[i].split(':');
    rec(f[0].slice(0, 64), f[2], 'sign.input line ' + i);
    var last = cases[cases.length - 1];
    if (last.sig !== f[3].slice(0, 128) || last.pub !== f[1])
      throw new Error('reference disagrees with sign.input line ' + i);
  });
  [0, 1, 2, 31, 32, 46, 47, 48, 49, 63, 64, 78, 79, 80, 81, 95, 96, 111, 112, 113, 127, 128, 129,
    174, 175, 176, 177, 191, 192, 207, 208, 209, 255, 256, 257, 300, 511, 640].forEach(function(len) {
    rec(rng.bytes(32).toString('hex'), rng.bytes(len).toString('hex'), 'seeded, ' + len + '-byte message');
  });
  // secrets whose hash has extreme clamped bytes are as good as random; a few fixed patterns
  ['00', 'ff', '80', '01'].forEach(function(b) {
    rec(new Array(33).join(b), rng.bytes(40).toString('hex'), 'secret of 32 x 0x' + b);
  });
  return cases;
}

// ------------------------------------------------------------- sign_<curve>.json
// EC#sign (ec/index.js:110-186) with the nonce supplied through options.k, so that
// (hash, d, k) -> (r, s, recoveryParam) is a pure function; rejected nonces (k <= 1,
// k >= n-1, r == 0, s == 0) are recorded as rejected: true.
function genSign(name) {
  var pc = elliptic.curves[name];
  var ec = new elliptic.ec(pc);
  var c = pc.curve;
  var NB = c.n.byteLength();
  var rng = new Prng('ellgpu-golden-v1:sign:' + name);
  var cases = [];
  var N = Math.ceil(COUNTS[name] / 2);
  function one(z, d, k, canonical) {
    var o = { z: Buffer.from(z).toString('hex'), d: hex(d, NB), k: Buffer.from(k).toString('hex'),
      canonical: canonical };
    var calls = 0;
    try {
      var sig = ec.sign(z, ec.keyFromPrivate(hex(d, NB), 'hex'), { canonical: canonical,
        k: function(iter) { calls++; if (iter > 0) throw new Error('REJECTED'); return new BN(k); } });
      o.r = hex(sig.r, NB); o.s = hex(sig.s, NB); o.recid = sig.recoveryParam;
    } catch (e) {
      if (e.message !== 'REJECTED') throw e;
      o.rejected = true;
    }
    cases.push(o);
  }
  for (var i = 0; i < N; i++) {
    var zlen = [32, 32, 48, 64, 20][i % 5];
    var z = rng.bytes(zlen);
    var d = rng.below(c.n.subn(1)).addn(1);
    var k = rng.bytes(NB);
    if (NB * 8 === c.n.bitLength() && i % 3 === 0) k = Buffer.from(rng.below(c.n).toArray('be', NB));
    one(z, d, k, (i & 1) === 1);
  }
  var z0 = rng.bytes(32);
  var d0 = rng.below(c.n.subn(1)).addn(1);
  [new BN(0), new BN(1), new BN(2), c.n.subn(2), c.n.subn(1), c.n.clone()].forEach(function(kv) {
    // nonce bytes are truncated by bits like a digest (_truncateToN(k, true)), so place the
    // value such that it survives the shift
    var shift = NB * 8 - c.n.bitLength();
    one(z0, d0, Buffer.from(kv.ushln(shift).toArray('be', NB)), true);
  });
  return cases;
}

// ------------------------------------------------- captured_<curve>.json
// Run the reference's own mocha suite with the hot-path prototypes wrapped.
function captureFromReferenceTests() {
  var cap = {};
  var LIMIT = 48;
  var byP = {};
  SHORT.concat(['ed25519', 'curve25519']).forEach(function(name) {
    byP[elliptic.curves[name].curve.p.toString(16) + ':' +
        elliptic.curves[name].curve.type] = name;
    cap[name] = { mul: [], muladd: [], verify: [] };
  });
  function nameOf(curve) { return byP[curve.p.toString(16) + ':' + curve.type]; }
  function seen(list, key) {
    if (list._keys === undefined)
      Object.defineProperty(list, '_keys', { value: {}, enumerable: false });
    if (list._keys[key]) return true;
    list._keys[key] = 1;
    return false;
  }
  function scalarHex(curve, k) {
    var L = curve.type === 'short' ? flen(curve) : 32;
    if (k.isNeg() || k.byteLength() > L) return null;    // outside the C ABI
    return hex(k, L);
  }
  function wrapPoint(proto) {
    var mul = proto.mul;
    proto.mul = function(k) {
      var res = mul.apply(this, arguments);
      try {
        var cn = nameOf(this.curve);
        if (cn && !this.isInfinity()) {
          var kb = scalarHex(this.curve, BN.isBN(k) ? k : new BN(k, 16));
          var a = affine(this.curve, this);
          if (kb !== null) {
            var o = { k: kb, px: a.x, r: affine(this.curve, res) };
            if (a.y !== undefined) o.py = a.y;
            var key = JSON.stringify(o);
            if (cap[cn].mul.length < LIMIT && !seen(cap[cn].mul, key))
              cap[cn].mul.push(o);
          }
        }
      } catch (e) { /* capture must never disturb the suite */ }
      return res;
    };
    ['mulAdd', 'jmulAdd'].forEach(function(fn) {
      var orig = proto[fn];
      if (!orig) return;
      proto[fn] = function(k1, p2, k2) {
        var res = orig.apply(this, arguments);
        try {
          var cn = nameOf(this.curve);
          if (cn && !this.isInfinity() && !p2.isInfinity()) {
            var a = affine(this.curve, this);
            var b = affine(this.curve, p2);
            var h1 = scalarHex(this.curve, k1);
            var h2 = scalarHex(this.curve, k2);
            if (h1 !== null && h2 !== null) {
              var o = { k1: h1, p1x: a.x, p1y: a.y, k2: h2, p2x: b.x,
                p2y: b.y, r: affine(this.curve, res) };
              var key = JSON.stringify(o);
              if (cap[cn].muladd.length < LIMIT && !seen(cap[cn].muladd, key))
                cap[cn].muladd.push(o);
            }
          }
        } catch (e) { /* ignore */ }
        return res;
      };
    });
  }
  wrapPoint(elliptic.curves.secp256k1.curve.g.constructor.prototype);
  wrapPoint(elliptic.curves.ed25519.curve.g.constructor.prototype);
  wrapPoint(elliptic.curves.curve25519.curve.g.constructor.prototype);

  var verify = elliptic.ec.prototype.verify;
  elliptic.ec.prototype.verify = function(msg, signature, key, enc, options) {
    var res = verify.apply(this, arguments);
    try {
      var cn = nameOf(this.curve);
      if (cn && this.curve.type === 'short') {
        if (typeof enc === 'object') { options = enc; enc = null; }
        var kk = this.keyFromPublic(key, enc);
        var sg = new ref.Signature(signature, 'hex');
        var z = new BN(msg, 16);
        // effective byte length exactly as _truncateToN derives it
        // (ec/index.js:82-96): BN/number -> value length, array-like ->
        // .length, anything else -> hex string length
        var zb;
        if (BN.isBN(msg) || typeof msg === 'number') zb = z.byteLength();
        else if (typeof msg === 'object') zb = msg.length;
        else zb = (msg.toString().length + 1) >>> 1;
        var NL = this.n.byteLength();
        if (!z.isNeg() && zb <= 66 && zb >= z.byteLength() && zb > 0 &&
            sg.r.byteLength() <= NL &&
            sg.s.byteLength() <= NL && !sg.r.isNeg() && !sg.s.isNeg()) {
          var L = flen(this.curve);
          var o = { z: hex(z, zb), r: hex(sg.r, NL),
            s: hex(sg.s, NL), qx: hex(kk.getPublic().getX(), L),
            qy: hex(kk.getPublic().getY(), L), ok: res };
          if (options && options.msgBitLength)
            o.msgBitLength = options.msgBitLength;
          var ks = JSON.stringify(o);
          if (cap[cn].verify.length < LIMIT && !seen(cap[cn].verify, ks))
            cap[cn].verify.push(o);
        }
      }
    } catch (e) { /* ignore */ }
    return res;
  };

  var stats = require('./run_ref_tests').run(ref, { quiet: true });
  return { cap: cap, stats: stats };
}

// --------------------------------------------------------------- main ----
// GOLDEN_ONLY=<regexp>: (re)write only the fixtures whose file name matches -- every generator
// seeds its own PRNG with its file's name, so a partial run reproduces exactly those files
var ONLY = process.env.GOLDEN_ONLY ? new RegExp(process.env.GOLDEN_ONLY) : null;
function write(name, obj) {
  if (ONLY && !ONLY.test(name)) return;
  var file = path.join(OUT, name);
  fs.writeFileSync(file, JSON.stringify(obj, null, 0)
    .replace(/\},\{/g, '},\n{') + '\n');
  console.log('wrote', file);
}

fs.mkdirSync(OUT, { recursive: true });
write('curves.json', dumpCurves());
SHORT.forEach(function(name) {
  write('mul_' + name + '.json', genShortMul(name));
  write('verify_' + name + '.json', genVerify(name));
  write('sign_' + name + '.json', genSign(name));
  write('recover_' + name + '.json', genRecover(name));
  write('signdet_' + name + '.json', genSignDet(name));
});
SHORT.concat(['ed25519']).forEach(function(name) {
  write('decompress_' + name + '.json', genDecompress(name));
});
write('fromx_ed25519.json', genEdFromX());
SHORT.concat(['ed25519']).forEach(function(name) {
  write('codec_' + name + '.json', genCodec(name));
});
SHORT.forEach(function(name) {
  write('wire_' + name + '.json', genWire(name));
});
write('der_fuzz_secp256k1.json', genDerFuzz());
SHORT.concat(['ed25519']).forEach(function(name) {
  write('add_' + name + '.json', genAdd(name));
});
SHORT.concat(['ed25519']).forEach(function(name) {
  write('offcurve_' + name + '.json', genOffCurve(name));
});
write('eddsa_verify_ed25519.json', genEddsa());
write('eddsa_sign_ed25519.json', genEddsaSign());
// public-API calls whose arguments are objects of the library (tools/api_forms.js): recipe + what
// the reference answers (a value or the message of the Error it throws)
(function() {
  var forms = require('./api_forms');
  var rng = new Prng('golden:api_forms');
  write('api_forms.json', forms.recipes(rng, elliptic).map(function(o) { o.want = forms.run(elliptic, o); return o; }));
})();
// inputs the reference trusts (tools/trusted_inputs.js): precomputed tables, endomorphism constants,
// degenerate curve equations -- recipe + what the reference answers
(function() {
  var trusted = require('./trusted_inputs');
  var rng = new Prng('golden:trusted_inputs');
  write('trusted_inputs.json', trusted.recipes(rng).map(function(o) { o.want = trusted.run(elliptic, o); return o; }));
})();
write('mul_ed25519.json', genEdwardsMul());
write('mul_curve25519.json', genMontMul());
if (ONLY && !ONLY.test('captured_')) process.exit(0);
var c = captureFromReferenceTests();
console.log('reference suite under capture:', JSON.stringify(c.stats));
if (c.stats.failed !== 0) throw new Error('reference suite failed under capture');
Object.keys(c.cap).forEach(function(name) {
  write('captured_' + name + '.json', c.cap[name]);
});
write('MANIFEST.json', {
  generator: 'tools/gen_golden.js',
  reference: 'indutny/elliptic ' + elliptic.version +
    ' (dist/elliptic.js, vendored bn.js 4.11.9)',
  node: process.version,
  reference_suite: c.stats,
});
