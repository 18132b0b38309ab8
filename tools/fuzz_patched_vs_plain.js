'use strict';
// Differential fuzz of the drop-in boundary: the SAME seeded sequence of public-API calls on an
// unpatched copy of the reference and on a copy patched by install() -- results (canonical affine
// coordinates, signatures, booleans) and exception MESSAGES must be identical.  Arguments are drawn
// to sit on the seams: scalars 0, 1, n - 1, n, n + 1, 2^bits - 1, wider than the curve, negative,
// numbers, hex strings; points that are G, fresh copies of G, -P, P itself twice, infinity, tabled
// points, off-curve points, non-canonical coordinates; messages as arrays, hex strings, BNs, empty,
// longer than n; signatures as objects, DER hex, DER arrays, corrupted; keys as points, hex, objects.
//   ELLGPU_LIB=<hostsim or real library> node tools/fuzz_patched_vs_plain.js [iterations] [seed]
var crypto = require('crypto');
var loader = require('./ref_loader');
var A = loader.load(), B = loader.load();
var plain = A.elliptic, patched = B.elliptic;
var eng = require('../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB });
var ITER = +(process.argv[2] || 400), SEED = process.argv[3] || 'fuzz-1';

function Prng(seed) { this.seed = seed; this.ctr = 0; }
Prng.prototype.bytes = function(n) {
  var out = [];
  while (out.length < n) {
    var h = crypto.createHash('sha256').update(this.seed + ':' + (this.ctr++)).digest();
    for (var i = 0; i < h.length && out.length < n; i++) out.push(h[i]);
  }
  return out;
};
Prng.prototype.int = function(m) { var b = this.bytes(4); return (((b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3]) >>> 0) % m; };
Prng.prototype.pick = function(a) { return a[this.int(a.length)]; };

function canon(v) {
  // a comparable rendering of whatever a call returned
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

var SHORT = ['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521'];
var rng = new Prng(SEED);
var stats = { calls: 0, threw: 0, byOp: {} };
var failures = [];

function scalar(L, c, BN) {
  var n = c.n, bits = c.p.bitLength();
  var k = rng.int(16);
  switch (k) {
    case 0: return new BN(0);
    case 1: return new BN(1);
    case 2: return n.subn(1);
    case 3: return n.clone();
    case 4: return n.addn(1);
    case 5: return new BN(1).ushln(8 * c.p.byteLength()).subn(1);
    case 6: return new BN(rng.bytes(c.p.byteLength() + 3));            // wider than the curve
    case 7: return new BN(rng.bytes(8)).neg();                        // negative
    case 8: return new BN(rng.bytes(16));
    case 9: return c.endo ? c.endo.lambda.clone() : new BN(2);
    case 10: return new BN(1).ushln(bits - 1);
    default: return new BN(rng.bytes(c.p.byteLength())).umod(n);
  }
}
function pointOn(lib, name, which, seedScalar) {
  // the same point on either library, built from the same recipe
  var c = lib.curves[name].curve;
  var BN = c.p.constructor;
  var P = c.g.mul(new BN(seedScalar, 16));
  switch (which) {
    case 0: return c.g;
    case 1: return c.type === 'mont' ? c.point(c.g.getX(), new BN(1)) : c.point(c.g.getX(), c.g.getY());
    case 2: return P;
    case 3: return c.type === 'short' ? P.neg() : P;
    case 4: return c.type === 'short' ? c.point(null, null) : P;
    case 5: if (c.type !== 'mont') { var T = c.point(P.getX(), P.getY()); T.precompute(c.n.bitLength() + 1); return T; } return P;
    case 6: return c.type === 'mont' ? c.point(new BN(seedScalar, 16).umod(c.p), new BN(1)) :
      c.point(new BN(seedScalar, 16).umod(c.p), new BN(seedScalar, 16).addn(7).umod(c.p));            // off the curve (w.h.p.)
    case 7: return c.type === 'short' ? c.point(P.getX().add(c.p), P.getY()) : P;                    // non-canonical x (reduced by toRed)
    default: return c.type === 'mont' ? P : c.point(P.getX(), P.getY());
  }
}

var context = '';
function both(op, fa, fb) {
  var a = run(fa), b = run(fb);
  stats.calls++;
  stats.byOp[op] = (stats.byOp[op] || 0) + 1;
  if (a[0] === 'e') stats.threw++;
  if (a !== b) failures.push({ op: op, args: context, reference: a.slice(0, 300), patched: b.slice(0, 300) });
}

for (var it = 0; it < ITER && failures.length < 5; it++) {
  var name = rng.pick(SHORT.concat(['secp256k1', 'p256', 'ed25519', 'ed25519', 'curve25519']));
  var ca = plain.curves[name].curve, cb = patched.curves[name].curve;
  var BNa = ca.p.constructor, BNb = cb.p.constructor;
  var kind = rng.int(10);
  var s1 = Buffer.from(rng.bytes(20)).toString('hex'), s2 = Buffer.from(rng.bytes(20)).toString('hex');
  var w1 = rng.int(9), w2 = rng.int(9);
  if (ca.type === 'edwards') { ca.g.precompute(ca.n.bitLength() + 1); cb.g.precompute(cb.n.bitLength() + 1); }
  (function() {
    var rs = rng.ctr;                                 // scalars drawn identically for both libraries
    function scal(c, BN) { var save = rng.ctr; rng.ctr = rs; var k = scalar(0, c, BN); rs = rng.ctr; rng.ctr = save; return k; }
    if (kind <= 2 || ca.type === 'mont') {
      var kA = scal(ca, BNa); rs -= (rng.ctr, 0);
      var kh = kA.toString(16);
      both(name + ' mul', function() { return pointOn(plain, name, w1, s1).mul(new BNa(kh, 16)); },
        function() { return pointOn(patched, name, w1, s1).mul(new BNb(kh, 16)); });
    } else if (kind <= 5) {
      var k1 = scal(ca, BNa).toString(16), k2 = scal(ca, BNa).toString(16);
      var jm = kind === 5 && ca.type === 'short';
      context = JSON.stringify({ w1: w1, s1: s1, k1: k1, w2: w2, s2: s2, k2: k2,
        gTablesRef: !!(ca.g.precomputed && ca.g.precomputed.naf), gTablesPatched: !!(cb.g.precomputed && cb.g.precomputed.naf) });
      both(name + (jm ? ' jmulAdd' : ' mulAdd'), function() {
        var P = pointOn(plain, name, w1, s1), Q = pointOn(plain, name, w2, s2);
        return jm ? P.jmulAdd(new BNa(k1, 16), Q, new BNa(k2, 16)) : P.mulAdd(new BNa(k1, 16), Q, new BNa(k2, 16));
      }, function() {
        var P = pointOn(patched, name, w1, s1), Q = pointOn(patched, name, w2, s2);
        return jm ? P.jmulAdd(new BNb(k1, 16), Q, new BNb(k2, 16)) : P.mulAdd(new BNb(k1, 16), Q, new BNb(k2, 16));
      });
    } else if (ca.type === 'short') {
      // ECDSA: sign with one library's key material on both, verify / recover variants
      var eca = new plain.ec(name), ecb = new patched.ec(name);
      var priv = new BNa(rng.bytes(ca.n.byteLength())).umod(ca.n.subn(1)).addn(1).toString(16);
      var mlen = rng.pick([0, 1, 20, 32, 32, 32, 48, 64, 70]);
      var mb = rng.bytes(mlen);
      var mform = rng.int(5);
      var msg = mform === 0 ? mb : mform === 1 ? Buffer.from(mb).toString('hex') : mform === 2 ? Buffer.from(mb) :
        mform === 3 ? mb.map(function(x, i) { return i === 0 ? x + 256 : x; }) : mb;
      var opts = rng.int(4) === 0 ? { canonical: true } : rng.int(6) === 0 ? { msgBitLength: rng.pick([0, 8, 260, -1, '256', 1.5]) } : undefined;
      both(name + ' sign', function() { return eca.sign(msg, priv, 'hex', opts); }, function() { return ecb.sign(msg, priv, 'hex', opts); });
      var sig;
      try { sig = eca.sign(mb, priv, 'hex'); } catch (e) { sig = null; }
      if (sig) {
        var form = rng.int(5);
        var sg = form === 0 ? { r: sig.r.toString(16), s: sig.s.toString(16) } : form === 1 ? sig.toDER('hex') : form === 2 ? sig.toDER() :
          form === 3 ? { r: sig.r.xor(new BNa(1)).toString(16), s: sig.s.toString(16) } : sig.toDER('hex').slice(0, -2);
        var pubHex = eca.keyFromPrivate(priv, 'hex').getPublic(rng.int(2) === 0, 'hex');
        var keyForm = rng.int(4);
        var keyA = keyForm === 0 ? pubHex : keyForm === 1 ? pointOn(plain, name, 6, s1) : keyForm === 2 ? '05' + pubHex.slice(2) : pubHex;
        var keyB = keyForm === 1 ? pointOn(patched, name, 6, s1) : keyA;
        var mv = rng.int(3) === 0 ? Buffer.from(mb).toString('hex') : mb;
        both(name + ' verify', function() { return eca.verify(mv, sg, keyA, 'hex'); }, function() { return ecb.verify(mv, sg, keyB, 'hex'); });
        // EC#verify itself is patched (one engine call): every form the reference takes for its
        // key (KeyPair with / without a public half, {x, y}, a Point, O), its signature, its
        // message (Buffer, Array, Uint8Array; any length) and options.msgBitLength
        var kf2 = rng.int(6);
        var xyA = eca.keyFromPrivate(priv, 'hex').getPublic();
        function key2(ec, lib, c) {
          return kf2 === 0 ? ec.keyFromPrivate(priv, 'hex') : kf2 === 1 ? ec.keyFromPublic(pubHex, 'hex') :
            kf2 === 2 ? c.point(null, null) : kf2 === 3 ? { x: xyA.getX().toString(16), y: xyA.getY().toString(16) } :
            kf2 === 4 ? pointOn(lib, name, rng.ctr % 7, s2) : ec.keyFromPrivate(priv, 'hex').getPublic();
        }
        var vopts = rng.int(3) === 0 ? { msgBitLength: rng.pick([8, 160, 255, 256, 260, 512, 0, -8, 100.5, '260', NaN]) } : undefined;
        var mf2 = rng.int(3);
        var mv2 = mf2 === 0 ? Buffer.from(mb) : mf2 === 1 ? Uint8Array.from(mb) : mb;
        var sgf = rng.int(2);
        var save2 = rng.ctr;
        both(name + ' verify (key / options forms)', function() { rng.ctr = save2; return eca.verify(mv2, sgf ? sig : sg, key2(eca, plain, ca), undefined, vopts); },
          function() {
            rng.ctr = save2;
            var sB = sgf ? { r: new BNb(sig.r.toString(16), 16), s: new BNb(sig.s.toString(16), 16) } : sg;
            return ecb.verify(mv2, sB, key2(ecb, patched, cb), undefined, vopts);
          });
        var j = rng.int(5);
        both(name + ' recoverPubKey', function() { return eca.recoverPubKey(mb, sg, j); }, function() { return ecb.recoverPubKey(mb, sg, j); });
        var xh = new BNa(rng.bytes(ca.p.byteLength())).toString(16);
        both(name + ' pointFromX', function() { return ca.pointFromX(xh, rng.ctr % 2 === 0); }, function() { return cb.pointFromX(xh, rng.ctr % 2 === 0); });
        both(name + ' derive', function() { return eca.keyFromPrivate(priv, 'hex').derive(eca.keyFromPublic(pubHex, 'hex').getPublic()); },
          function() { return ecb.keyFromPrivate(priv, 'hex').derive(ecb.keyFromPublic(pubHex, 'hex').getPublic()); });
      }
    } else if (ca.type === 'edwards') {
      var eda = new plain.eddsa('ed25519'), edb = new patched.eddsa('ed25519');
      var secLen = rng.pick([32, 32, 32, 16, 40]);
      var secret = Buffer.from(rng.bytes(secLen)).toString('hex');
      var m2 = rng.bytes(rng.pick([0, 1, 3, 32, 100]));
      both('eddsa sign', function() { return eda.sign(m2, secret).toHex(); }, function() { return edb.sign(m2, secret).toHex(); });
      var sigh;
      try { sigh = eda.sign(m2, secret).toHex(); } catch (e) { sigh = null; }
      if (sigh) {
        var pubh = eda.keyFromSecret(secret).getPublic('hex');
        var tw = rng.int(4);
        var sgh = tw === 1 ? (sigh.slice(0, 10) + (sigh[10] === 'a' ? 'b' : 'a') + sigh.slice(11)) : sigh;
        var ph = tw === 2 ? ('ff' + pubh.slice(2)) : pubh;
        both('eddsa verify', function() { return eda.verify(m2, sgh, ph); }, function() { return edb.verify(m2, sgh, ph); });
        var yh = new BNa(rng.bytes(32)).toString(16);
        both('ed pointFromY', function() { return ca.pointFromY(yh, false); }, function() { return cb.pointFromY(yh, false); });
      }
    }
  })();
}
// user-defined curves (run-time modulus kernels): the reference's own test curves and three others,
// as tools/gen_golden_custom.js defines them -- Point#mul / mulAdd / jmulAdd / add with seam scalars,
// infinity, off-curve points; scalars as numbers and hex strings on a preset (Point#mul converts them)
(function() {
  var fs = require('fs'), path = require('path');
  var specs = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'custom_short.json')));
  var eds = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', 'custom_edwards.json')));
  function mkShort(lib, sp) {
    return new lib.curve.short({ p: sp.p, a: sp.a, b: sp.b, n: sp.n, g: [sp.g.x, sp.g.y] });
  }
  function mkEd(lib, sp) {
    return new lib.curve.edwards({ p: sp.p, a: sp.a, c: '1', d: sp.d, n: sp.n || null, g: [sp.g.x, sp.g.y] });
  }
  var made = [];
  specs.forEach(function(sp) { made.push({ name: sp.name, a: mkShort(plain, sp), b: mkShort(patched, sp), type: 'short' }); });
  eds.forEach(function(sp) { made.push({ name: sp.name, a: mkEd(plain, sp), b: mkEd(patched, sp), type: 'edwards' }); });
  for (var it = 0; it < Math.ceil(ITER / 4) && failures.length < 5; it++) {
    var m = rng.pick(made);
    var BNa = m.a.p.constructor, BNb = m.b.p.constructor;
    var nn = m.a.n || m.a.p;
    var ks = [new BNa(0), new BNa(1), nn.subn(1), nn.clone(), nn.addn(1), new BNa(rng.bytes(m.a.p.byteLength())),
      new BNa(rng.bytes(m.a.p.byteLength())).umod(nn), new BNa(rng.bytes(40)), new BNa(rng.bytes(4)).neg()];
    var k1 = rng.pick(ks).toString(16), k2 = rng.pick(ks).toString(16);
    var d1 = new BNa(rng.bytes(8)).toString(16), d2 = new BNa(rng.bytes(8)).toString(16);
    var off = rng.int(6) === 0;
    function pt(c, BN, d) {
      var P = c.g.mul(new BN(d, 16));
      if (c.type === 'edwards') return off ? c.point(P.getX().addn(1).umod(c.p), P.getY()) : c.point(P.getX(), P.getY());
      if (P.isInfinity()) return P;
      return off ? c.point(P.getX(), P.getY().addn(1).umod(c.p)) : P;
    }
    var op = rng.int(4);
    if (op === 0) both(m.name + ' mul', function() { return pt(m.a, BNa, d1).mul(new BNa(k1, 16)); }, function() { return pt(m.b, BNb, d1).mul(new BNb(k1, 16)); });
    else if (op === 1 && m.type === 'short') both(m.name + ' mulAdd', function() { return pt(m.a, BNa, d1).mulAdd(new BNa(k1, 16), pt(m.a, BNa, d2), new BNa(k2, 16)); },
      function() { return pt(m.b, BNb, d1).mulAdd(new BNb(k1, 16), pt(m.b, BNb, d2), new BNb(k2, 16)); });
    else if (op === 2 && m.type === 'short') both(m.name + ' jmulAdd', function() { return pt(m.a, BNa, d1).jmulAdd(new BNa(k1, 16), pt(m.a, BNa, d2), new BNa(k2, 16)); },
      function() { return pt(m.b, BNb, d1).jmulAdd(new BNb(k1, 16), pt(m.b, BNb, d2), new BNb(k2, 16)); });
    else both(m.name + ' add', function() { return pt(m.a, BNa, d1).add(pt(m.a, BNa, d2)); }, function() { return pt(m.b, BNb, d1).add(pt(m.b, BNb, d2)); });
  }
  // Point#mul with a number / a hex string (short.js:422: k = new BN(k, 16))
  ['secp256k1', 'p256', 'ed25519'].forEach(function(name) {
    [7, 0, 65537, 'ff', '0', 'deadbeefdeadbeefdeadbeefdeadbeefdeadbeefdeadbeefdeadbeefdeadbeef01'].forEach(function(k) {
      both(name + ' mul(non-BN)', function() { return pointOn(plain, name, 8, 'abcdef').mul(k); }, function() { return pointOn(patched, name, 8, 'abcdef').mul(k); });
      both(name + ' g.mul(non-BN)', function() { return plain.curves[name].curve.g.mul(k); }, function() { return patched.curves[name].curve.g.mul(k); });
    });
  });
})();
// ---- the forms of the public API that carry OBJECTS (round-4 review: three mismatches hid here) ----
// EDDSA#verify computes with the point objects it is given -- the key's, the signature's R -- and
// with a BN S, whatever their encodings say (eddsa/index.js:52-63, eddsa/key.js:20-23,
// eddsa/signature.js:33-38): off-curve twins that encode like the true point, points with Z != 1
// or a wrong T, a negative S, S + n, an `Rencoded` / `Sencoded` that belongs to another signature.
(function() {
  function build(L, o) {
    var ed = new L.eddsa('ed25519'), c = ed.curve, BN = c.p.constructor;
    var sig = ed.sign(o.msg, o.secret), other = ed.sign(o.msg.concat([ 7 ]), o.secret);
    var A = ed.keyFromSecret(o.secret).pub();
    function twin(P) { return c.point(P.getX().addn(2).umod(c.p), P.getY()); }              // same encoding, off the curve
    function scaled(P) { var z = new BN(o.z, 16).toRed(c.red); return c.point(P.x.redMul(z), P.y.redMul(z), z, P.t.redMul(z)); }
    function badT(P) { return c.point(P.getX(), P.getY(), null, P.t.redAdd(c.one)); }
    function fresh(P) { return c.point(P.getX(), P.getY()); }
    var pub;
    switch (o.pubForm) {
      case 0: pub = ed.keyFromSecret(o.secret).getPublic('hex'); break;
      case 1: pub = ed.keyFromSecret(o.secret).getPublic(); break;
      case 2: pub = fresh(A); break;
      case 3: pub = twin(A); break;
      case 4: pub = scaled(A); break;
      case 5: pub = badT(A); break;
      case 6: pub = ed.keyFromPublic(fresh(A)); break;
      case 7: pub = ed.keyFromPublic(twin(A)); break;
      case 8: pub = ed.keyFromPublic(ed.keyFromSecret(o.secret).getPublic()); break;
      case 9: pub = ed.keyFromSecret(o.secret); break;
      case 10: pub = ed.keyFromPublic(fresh(A)); pub.pubBytes(); pub.pub(); break;           // caches filled
      case 11: pub = new L.eddsa('ed25519').keyFromPublic(fresh(A)); break;                  // another EDDSA instance
      default: pub = Buffer.from(ed.keyFromSecret(o.secret).getPublic());                    // a Buffer is no Array
    }
    var R = sig.R(), S = sig.S(), sg;
    switch (o.sigForm) {
      case 0: sg = sig.toHex(); break;
      case 1: sg = sig.toBytes(); break;
      case 2: sg = { R: sig.Rencoded(), S: sig.Sencoded() }; break;
      case 3: sg = { R: fresh(R), S: S.clone() }; break;
      case 4: sg = { R: twin(R), S: S.clone() }; break;
      case 5: sg = { R: fresh(R), S: S.neg() }; break;
      case 6: sg = { R: fresh(R), S: S.add(c.n) }; break;
      case 7: sg = { R: scaled(R), S: S.clone() }; break;
      case 8: sg = { R: badT(R), S: S.clone() }; break;
      case 9: sg = { R: fresh(other.R()), S: S.clone(), Rencoded: sig.Rencoded() }; break;
      case 10: sg = { R: sig.Rencoded(), S: other.S(), Sencoded: sig.Sencoded() }; break;
      case 11: sg = sig; break;
      case 12: sg = ed.makeSignature({ R: twin(R), S: S.clone() }); break;
      case 13: sg = { R: fresh(R), S: sig.Sencoded() }; break;
      case 14: sg = { R: sig.Rencoded(), S: S.clone() }; break;
      case 15: sg = { R: fresh(R), S: new BN(1).ushln(300) }; break;
      case 16: sg = { R: twin(R), S: S.clone(), Rencoded: sig.Rencoded() }; break;
      default: sg = new L.eddsa('ed25519').makeSignature(sig.toHex());                       // a Signature of another instance
    }
    var msg = o.msgForm === 0 ? o.msg : o.msgForm === 1 ? Buffer.from(o.msg).toString('hex') :
      o.msgForm === 2 ? o.msg.map(function(x, i) { return i === 1 ? x + 256 : x; }) : Buffer.from(o.msg);
    return { ed: ed, msg: msg, sig: sg, pub: pub };
  }
  for (var it = 0; it < Math.ceil(ITER / 2) && failures.length < 5; it++) {
    var o = { secret: Buffer.from(rng.bytes(32)).toString('hex'), msg: rng.bytes(rng.pick([ 2, 3, 32, 64 ])),
      z: Buffer.from(rng.bytes(31)).toString('hex') + '01', pubForm: rng.int(13), sigForm: rng.int(18), msgForm: rng.int(8) > 5 ? rng.int(4) : 0 };
    context = JSON.stringify(o);
    both('eddsa verify (object forms)', function() { var b = build(plain, o); return b.ed.verify(b.msg, b.sig, b.pub); },
      function() { var b = build(patched, o); return b.ed.verify(b.msg, b.sig, b.pub); });
    // ... and twice on the same objects (the reference caches decoded points / encodings on them)
    both('eddsa verify (object forms, same objects twice)', function() { var b = build(plain, o); b.ed.verify(b.msg, b.sig, b.pub); return b.ed.verify(b.msg, b.sig, b.pub); },
      function() { var b = build(patched, o); b.ed.verify(b.msg, b.sig, b.pub); return b.ed.verify(b.msg, b.sig, b.pub); });
    if (it % 4 === 0) {
      var sform = rng.int(4);
      both('eddsa sign (secret / message forms)', function() {
        var ed = new plain.eddsa('ed25519');
        var sec = sform === 0 ? o.secret : sform === 1 ? ed.keyFromSecret(o.secret) : sform === 2 ? Buffer.from(o.secret, 'hex') :
          Array.prototype.slice.call(Buffer.from(o.secret, 'hex')).map(function(x, i) { return i === 3 ? x + 256 : x; });
        return ed.sign(build(plain, o).msg, sec).toHex();
      }, function() {
        var ed = new patched.eddsa('ed25519');
        var sec = sform === 0 ? o.secret : sform === 1 ? ed.keyFromSecret(o.secret) : sform === 2 ? Buffer.from(o.secret, 'hex') :
          Array.prototype.slice.call(Buffer.from(o.secret, 'hex')).map(function(x, i) { return i === 3 ? x + 256 : x; });
        return ed.sign(build(patched, o).msg, sec).toHex();
      });
      // Edwards Point#mul / mulAdd on points whose coordinates are not what a decoder makes
      var kk = Buffer.from(rng.bytes(32)).toString('hex'), pf = rng.int(4);
      both('ed25519 mul (Z != 1 / wrong T / off-curve twin)', function() {
        var c = plain.curves.ed25519.curve, BN = c.p.constructor, A = c.g.mul(new BN(o.secret, 16));
        var z = new BN(o.z, 16).toRed(c.red);
        var P = pf === 0 ? c.point(A.x.redMul(z), A.y.redMul(z), z, A.t.redMul(z)) : pf === 1 ? c.point(A.getX(), A.getY(), null, A.t.redAdd(c.one)) :
          pf === 2 ? c.point(A.getX().addn(2).umod(c.p), A.getY()) : c.point(A.getX(), A.getY(), new BN(1));
        return P.mul(new BN(kk, 16).umod(c.n));
      }, function() {
        var c = patched.curves.ed25519.curve, BN = c.p.constructor, A = c.g.mul(new BN(o.secret, 16));
        A = c.point(A.getX(), A.getY());
        var z = new BN(o.z, 16).toRed(c.red);
        var P = pf === 0 ? c.point(A.x.redMul(z), A.y.redMul(z), z, A.t.redMul(z)) : pf === 1 ? c.point(A.getX(), A.getY(), null, A.t.redAdd(c.one)) :
          pf === 2 ? c.point(A.getX().addn(2).umod(c.p), A.getY()) : c.point(A.getX(), A.getY(), new BN(1));
        return P.mul(new BN(kk, 16).umod(c.n));
      });
    }
  }
})();
// EC#sign / EC#verify with digests of NB - 1 .. NB + 4 bytes and an explicit options.msgBitLength
// on every curve: the reference writes the truncated digest on n.byteLength() bytes
// (ec/index.js:133-139) and throws when it does not fit -- p521, 67 / 68 bytes
(function() {
  SHORT.forEach(function(name) {
    var eca = new plain.ec(name), ecb = new patched.ec(name);
    var NB = eca.n.byteLength(), bits = eca.n.bitLength();
    var priv = Buffer.from(rng.bytes(NB - 1)).toString('hex');
    for (var len = NB - 1; len <= NB + 4 && failures.length < 5; len++) {
      var mb = rng.bytes(len);
      mb[0] |= 0x80;                                  // a digest that really is `len` bytes wide
      var full;
      try { full = eca.sign(mb, priv, 'hex'); } catch (e) { full = null; }
      [ undefined, 1, 8, bits - 1, bits, bits + 1, 8 * len - 1, 8 * len, 8 * len + 8 ].forEach(function(mbl) {
        var opts = mbl === undefined ? undefined : { msgBitLength: mbl };
        context = JSON.stringify({ curve: name, len: len, msgBitLength: mbl });
        both(name + ' sign (digest length x msgBitLength)', function() { return eca.sign(mb, priv, 'hex', opts); }, function() { return ecb.sign(mb, priv, 'hex', opts); });
        var sg;
        try { sg = eca.sign(mb, priv, 'hex', opts); } catch (e) { sg = full; }
        if (!sg) return;
        var sh = { r: sg.r.toString(16), s: sg.s.toString(16) };
        var pubHex = eca.keyFromPrivate(priv, 'hex').getPublic('hex');
        both(name + ' verify (digest length x msgBitLength)', function() { return eca.verify(mb, sh, pubHex, 'hex', opts); }, function() { return ecb.verify(mb, sh, pubHex, 'hex', opts); });
      });
    }
  });
})();
// A KeyPair is used as it is (ec/key.js:23-24, 31-32), whichever EC instance made it: its private
// half was reduced by ITS curve's order, its public half lives on ITS curve
(function() {
  [ [ 'secp256k1', 'p256' ], [ 'p256', 'secp256k1' ], [ 'p384', 'p256' ], [ 'p224', 'p192' ], [ 'p521', 'p384' ], [ 'p256', 'p256' ] ].forEach(function(pr) {
    function mk(L) { return { from: new L.ec(pr[0]), to: new L.ec(pr[1]) }; }
    var a = mk(plain), b = mk(patched);
    var n2 = a.to.n;
    [ '07', n2.addn(5).toString(16), n2.subn(1).toString(16), n2.toString(16), Buffer.from(rng.bytes(a.from.n.byteLength())).toString('hex') ].forEach(function(ph) {
      if (failures.length >= 5) return;
      var msg = rng.bytes(32);
      context = JSON.stringify({ from: pr[0], to: pr[1], priv: ph });
      both('sign with a KeyPair of another EC instance', function() { return a.to.sign(msg, a.from.keyFromPrivate(ph, 'hex')); },
        function() { return b.to.sign(msg, b.from.keyFromPrivate(ph, 'hex')); });
      both('KeyPair#sign of another EC instance', function() { return a.from.keyFromPrivate(ph, 'hex').sign(msg); },
        function() { return b.from.keyFromPrivate(ph, 'hex').sign(msg); });
      var sg;
      try { sg = a.to.sign(msg, '0b', 'hex'); } catch (e) { return; }
      var sh = { r: sg.r.toString(16), s: sg.s.toString(16) };
      both('verify with a KeyPair of another EC instance', function() { return a.to.verify(msg, sh, a.from.keyFromPrivate(ph, 'hex')); },
        function() { return b.to.verify(msg, sh, b.from.keyFromPrivate(ph, 'hex')); });
      both('verify with a public-only KeyPair of another EC instance', function() { return a.to.verify(msg, sh, a.from.keyFromPublic(a.from.keyFromPrivate(ph, 'hex').getPublic('hex'), 'hex')); },
        function() { return b.to.verify(msg, sh, b.from.keyFromPublic(b.from.keyFromPrivate(ph, 'hex').getPublic('hex'), 'hex')); });
      both('derive across EC instances', function() { return a.to.keyFromPrivate('0d', 'hex').derive(a.from.keyFromPrivate(ph, 'hex').getPublic()); },
        function() { return b.to.keyFromPrivate('0d', 'hex').derive(b.from.keyFromPrivate(ph, 'hex').getPublic()); });
      both('derive with a private key of another EC instance', function() { return a.from.keyFromPrivate(ph, 'hex').derive(a.from.keyFromPrivate('11', 'hex').getPublic()); },
        function() { return b.from.keyFromPrivate(ph, 'hex').derive(b.from.keyFromPrivate('11', 'hex').getPublic()); });
    });
  });
})();
// Inputs the reference trusts (tools/trusted_inputs.js): points carrying precomputed tables -- right
// ones and tampered ones -- the curve's own G with such tables, endomorphism constants given in the
// curve's conf, toy curves over F_5 .. F_13 with any coefficients (singular cubics, Edwards curves
// without a complete addition law included), drawn at random
(function() {
  var T = require('./trusted_inputs');
  function hex(n) { return Buffer.from(rng.bytes(n)).toString('hex'); }
  var draws = Math.max(12, ITER >> 3);
  for (var i = 0; i < draws && failures.length < 5; i++) {
    var o, kind = rng.int(12);
    if (kind >= 10) {
      // ... and the same objects changed AFTER their first use (a remembered verdict must not outlive the change)
      o = { op: 'mutate', curve: rng.pick([ 'secp256k1', 'secp256k1', 'p256', 'p224', 'ed25519' ]), subject: rng.pick([ 'P', 'G' ]),
        mutation: rng.pick(T.MUTATE), table: rng.pick([ 'naf', 'doubles' ]), at: rng.int(250), mult: 2 + rng.int(9),
        k: hex(1 + rng.int(32)), k2: hex(1 + rng.int(32)), d: hex(1 + rng.int(24)), msg: rng.bytes(32) };
      if (o.mutation === 'self-x' && o.subject === 'G') o.subject = 'P';
    } else if (kind < 6) {
      var curve = rng.pick([ 'secp256k1', 'secp256k1', 'p256', 'p224', 'p384', 'ed25519' ]);
      o = { op: 'tables', curve: curve, mult: 1 + rng.int(9), tamper: rng.pick(T.TAMPER), at: rng.int(250),
        call: rng.pick(curve === 'ed25519' ? [ 'mul', 'mulAdd', 'mulAddRev' ] : [ 'mul', 'mulAdd', 'mulAddRev', 'jmulAdd', 'derive', 'verify' ]),
        k: hex(1 + rng.int(plain.curves[curve].curve.n.byteLength())), k2: hex(1 + rng.int(32)), msg: rng.bytes(32) };
    } else if (kind < 7) {
      o = { op: 'g-tables', curve: rng.pick([ 'secp256k1', 'p256', 'ed25519' ]), tamper: rng.pick(T.TAMPER), at: rng.int(250),
        k: hex(20), k2: hex(31), msg: rng.bytes(32) };
    } else if (kind < 8 && rng.int(2)) {
      o = { op: 'foreign-red', curve: rng.pick([ 'secp256k1', 'p256', 'p224', 'ed25519', 'curve25519' ]), mult: hex(12), k: hex(1 + rng.int(32)),
        msg: rng.bytes(32), secret: hex(32) };
    } else if (kind < 8) {
      o = { op: 'endo', variant: rng.pick(T.ENDO), mult: hex(8), k: hex(1 + rng.int(32)), k2: hex(1 + rng.int(32)) };
    } else {
      var P = rng.pick([ 5, 7, 7, 11, 13 ]);
      o = { op: 'toy', type: rng.pick([ 'short', 'edwards' ]), p: P, a: rng.int(P), b: rng.int(P), big: hex(32) };
      if (o.type === 'edwards' && (o.a === 0 || o.b === 0 || o.a === o.b)) { o.a = 1; o.b = 2; }
    }
    context = JSON.stringify(o).slice(0, 400);
    (function(o) {
      both('trusted inputs: ' + o.op, function() { return T.run(plain, o); }, function() { return T.run(patched, o); });
    })(o);
  }
})();
// uninstall() puts the reference's own functions back: every method install() replaces must read
// exactly like the unpatched library's again
(function() {
  eng.uninstall();
  var pairs = [
    ['base._fixedNafMul', plain.curve.base.prototype._fixedNafMul, patched.curve.base.prototype._fixedNafMul],
    ['base._wnafMul', plain.curve.base.prototype._wnafMul, patched.curve.base.prototype._wnafMul],
    ['base._wnafMulAdd', plain.curve.base.prototype._wnafMulAdd, patched.curve.base.prototype._wnafMulAdd],
    ['short._endoWnafMulAdd', plain.curve.short.prototype._endoWnafMulAdd, patched.curve.short.prototype._endoWnafMulAdd],
    ['short.pointFromX', plain.curve.short.prototype.pointFromX, patched.curve.short.prototype.pointFromX],
    ['edwards.pointFromY', plain.curve.edwards.prototype.pointFromY, patched.curve.edwards.prototype.pointFromY],
    ['edwards.pointFromX', plain.curve.edwards.prototype.pointFromX, patched.curve.edwards.prototype.pointFromX],
    ['ec.sign', plain.ec.prototype.sign, patched.ec.prototype.sign],
    ['ec.recoverPubKey', plain.ec.prototype.recoverPubKey, patched.ec.prototype.recoverPubKey],
    ['ec.verify', plain.ec.prototype.verify, patched.ec.prototype.verify],
    ['eddsa.verify', plain.eddsa.prototype.verify, patched.eddsa.prototype.verify],
    ['eddsa.sign', plain.eddsa.prototype.sign, patched.eddsa.prototype.sign],
    ['mont Point#mul', plain.curves.curve25519.curve.g.constructor.prototype.mul, patched.curves.curve25519.curve.g.constructor.prototype.mul],
    ['KeyPair#derive', new plain.ec('p192').keyFromPrivate('01', 'hex').derive, new patched.ec('p192').keyFromPrivate('01', 'hex').derive],
  ];
  pairs.forEach(function(q) {
    if (String(q[1]) !== String(q[2])) failures.push({ op: 'uninstall', args: q[0], reference: String(q[1]).slice(0, 80), patched: String(q[2]).slice(0, 80) });
  });
  // ... and the library works without the engine afterwards
  var e1 = new plain.ec('secp256k1'), e2 = new patched.ec('secp256k1');
  var k1 = e1.keyFromPrivate('0123456789abcdef', 'hex'), k2 = e2.keyFromPrivate('0123456789abcdef', 'hex');
  var before = eng.stats.gpuCalls;
  both('after uninstall: sign + verify', function() { var s = e1.sign([1, 2, 3], k1); return [s, e1.verify([1, 2, 3], s, k1)]; },
    function() { var s = e2.sign([1, 2, 3], k2); return [s, e2.verify([1, 2, 3], s, k2)]; });
  if (eng.stats.gpuCalls !== before) failures.push({ op: 'uninstall', args: 'engine still called', reference: '', patched: '' });
})();
if (failures.length) { console.log(JSON.stringify({ ok: false, seed: SEED, failures: failures }, null, 1)); process.exit(1); }
console.log(JSON.stringify({ ok: true, seed: SEED, calls: stats.calls, reference_threw: stats.threw, by_op: stats.byOp, engine: eng.stats }));
if (!process.env.ELLGPU_NATURAL_EXIT) process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
