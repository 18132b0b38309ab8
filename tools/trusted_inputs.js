'use strict';
// Inputs the reference TRUSTS.  Its ladders read a point's precomputed tables
// (lib/elliptic/curve/base.js:52-84 doubles, :96-253 naf, short.js:282-310 beta), the constants of
// the GLV endomorphism (short.js:28-75: conf.beta / conf.lambda / conf.basis are taken as given,
// :168-249 split and ladder) and the curve equation itself as they are handed in.  When those are
// not what precompute() / _getEndomorphism() would have made of the curve -- tables through
// curve.pointFromJSON, short.js:328-355; an order n smaller than the group; a singular cubic; an
// Edwards curve whose addition law is not complete -- the reference still answers, with a value
// that is not k * P.  install() answers from (x, y) and k alone, so it must leave every such call
// to the reference's own code (elliptic_amd/js/index.js: tablesOK, endoOK, customDomain).
// A case is a RECIPE (plain JSON); run(lib, recipe) builds the objects inside `lib` and returns a
// comparable rendering of the result ('v:...' or 'e:<message>').
//   tools/gen_golden.js             runs every recipe on the unpatched reference -> tests/golden/trusted_inputs.json
//   tools/check_patched_results.js  replays them through install() and compares
//   tools/fuzz_patched_vs_plain.js  draws random recipes of the same families
var crypto = require('crypto');

function pointStr(v) {
  if (v.isInfinity() && v.curve.type !== 'edwards') return 'O';
  if (v.curve.type === 'mont') return 'x=' + v.getX().toString(16);
  if (v.curve.type === 'edwards') {
    var q = v.curve.point(v.x, v.y, v.z, v.t);
    return '(' + q.getX().toString(16) + ',' + q.getY().toString(16) + ')';
  }
  if (v.type === 'jacobian') v = v.toP();
  return '(' + v.getX().toString(16) + ',' + v.getY().toString(16) + ')';
}
function str(v) {
  if (v === null || v === undefined || typeof v !== 'object') return String(v);
  if (Array.isArray(v)) return '[' + v.map(str).join(';') + ']';
  if (v.r && v.s && v.recoveryParam !== undefined) return 'sig(' + v.r.toString(16) + ',' + v.s.toString(16) + ',' + v.recoveryParam + ')';
  if (typeof v.isInfinity === 'function') return pointStr(v);
  if (v.constructor && v.constructor.name === 'BN') return 'bn' + v.toString(16);
  if (typeof v.toHex === 'function') return 'hex' + v.toHex();
  return JSON.stringify(v);
}
function render(f) { try { return 'v:' + str(f()); } catch (e) { return 'e:' + String(e && e.message); } }
function digest(list) { return list.length + ':' + crypto.createHash('sha256').update(list.join('|')).digest('hex').slice(0, 32); }

// ---- a point with tables of its own, right or wrong ------------------------------------------------
var TAMPER = [ 'none', 'json', 'doubles', 'doubles-last', 'naf', 'naf-unused', 'naf-short', 'naf-only', 'naf-only-bad',
  'wnd-smaller', 'wnd-larger', 'wnd-zero', 'step', 'beta', 'beta-tables', 'foreign-entry', 'unreduced-entry', 'null-tables' ];
function tabled(L, c, mult, tamper, at) {
  var BN = c.p.constructor;
  var base = c.g.mul(new BN(mult));
  var P = c.point(base.getX(), base.getY());
  if (tamper === 'null-tables') return P;
  P.precompute(c.n.bitLength() + 1);
  var pre = P.precomputed, wrong = c.g.mul(new BN(3));
  wrong = c.point(wrong.getX(), wrong.getY());
  switch (tamper) {
    case 'none': break;
    case 'json': P = c.pointFromJSON(JSON.parse(JSON.stringify(P.toJSON()))); break;
    case 'doubles': pre.doubles.points[1 + at % (pre.doubles.points.length - 1)] = wrong; break;
    case 'doubles-last': pre.doubles.points[pre.doubles.points.length - 1] = wrong; break;
    case 'naf': pre.naf.points[1 + at % 127] = wrong; break;
    case 'naf-unused': pre.naf.points[128 + at % 127] = wrong; break;        // past the reach of a width-8 digit
    case 'naf-short': pre.naf.points.length = 100; break;
    case 'naf-only': pre.doubles = null; break;
    case 'naf-only-bad': pre.doubles = null; pre.naf.points[1 + at % 127] = wrong; break;
    case 'wnd-smaller': pre.naf.wnd = 5; break;
    case 'wnd-larger': pre.naf.wnd = 9; break;
    case 'wnd-zero': pre.naf.wnd = 0; break;
    case 'step': pre.doubles.step = 3; break;
    case 'beta': if (pre.beta) pre.beta = wrong; else pre.naf.points[3] = wrong; break;
    case 'beta-tables': if (pre.beta) pre.beta.precomputed.naf.points[2] = wrong; else pre.naf.points[3] = wrong; break;
    case 'foreign-entry': {
      var c2 = c.type === 'short' ? new L.curve.short({ p: c.p.toString(16), a: c.a.fromRed().toString(16), b: c.b.fromRed().toString(16) }) :
        new L.curve.edwards({ p: c.p.toString(16), a: c.a.fromRed().toString(16), c: '1', d: c.d.fromRed().toString(16) });
      var e = pre.naf.points[2];
      pre.naf.points[2] = c2.point(e.getX(), e.getY());
      break;
    }
    case 'unreduced-entry': {
      var en = pre.naf.points[2];
      if (c.type === 'short') pre.naf.points[2] = c.point(en.getX().add(c.p), en.getY(), true);
      else pre.naf.points[2] = c.point(en.getX().add(c.p).forceRed(c.red), en.getY().toRed(c.red), c.one, en.t);
      break;
    }
    default: throw new Error('unknown tamper ' + tamper);
  }
  return P;
}

// a second curve object with a preset's parameters (so that its G can be given tables of its own
// without touching the library's shared preset): type, field, coefficients, generator, order
function cloneCurve(L, name, extra) {
  var c = L.curves[name].curve, conf;
  if (c.type === 'short') {
    conf = { p: c.p.toString(16), a: c.a.fromRed().toString(16), b: c.b.fromRed().toString(16), n: c.n.toString(16),
      g: [ c.g.getX().toString(16), c.g.getY().toString(16) ] };
  } else {
    var g = c.point(c.g.x, c.g.y, c.g.z, c.g.t);
    conf = { p: c.p.toString(16), a: c.a.fromRed().toString(16), c: '1', d: c.d.fromRed().toString(16), n: c.n.toString(16),
      g: [ g.getX().toString(16), g.getY().toString(16) ] };
  }
  Object.keys(extra || {}).forEach(function(k) { conf[k] = extra[k]; });
  return c.type === 'short' ? new L.curve.short(conf) : new L.curve.edwards(conf);
}

// an EC instance over a curve OBJECT (ec/index.js:24-45 reads options.curve.curve / .g / .n and a hash)
function ecOn(L, c, name) {
  return new L.ec({ curve: { curve: c, g: c.g, n: c.n, hash: L.curves[name].hash } });
}

var ENDO = [ 'auto', 'given', 'beta-other', 'lambda-other', 'basis-off', 'basis-other', 'basis-swapped' ];
function endoCurve(L, variant) {
  var c = L.curves.secp256k1.curve, e = c.endo, BN = c.p.constructor;
  var one = new BN(1);
  var beta2 = e.beta.redSqr().fromRed(), lambda2 = e.lambda.sqr().umod(c.n);
  function vec(v) { return { a: v.a.toString(16), b: v.b.toString(16) }; }
  var extra = {};
  switch (variant) {
    case 'auto': break;
    case 'given': extra = { beta: e.beta.fromRed().toString(16), lambda: e.lambda.toString(16), basis: e.basis.map(vec) }; break;
    case 'beta-other': extra = { beta: beta2.toString(16), lambda: e.lambda.toString(16), basis: e.basis.map(vec) }; break;
    case 'lambda-other': extra = { beta: e.beta.fromRed().toString(16), lambda: lambda2.toString(16), basis: e.basis.map(vec) }; break;
    case 'basis-off': extra = { beta: e.beta.fromRed().toString(16), lambda: e.lambda.toString(16),
      basis: [ vec({ a: e.basis[0].a.add(one), b: e.basis[0].b }), vec(e.basis[1]) ] }; break;
    case 'basis-other': extra = { beta: e.beta.fromRed().toString(16), lambda: e.lambda.toString(16),
      basis: [ vec(e.basis[0]), vec({ a: e.basis[0].a.add(e.basis[1].a), b: e.basis[0].b.add(e.basis[1].b) }) ] }; break;
    case 'basis-swapped': extra = { beta: e.beta.fromRed().toString(16), lambda: e.lambda.toString(16),
      basis: [ vec(e.basis[1]), vec(e.basis[0]) ] }; break;
    default: throw new Error('unknown endo variant ' + variant);
  }
  return cloneCurve(L, 'secp256k1', extra);
}

function allPoints(L, c, P, f) {
  var BN = c.p.constructor, pts = [];
  for (var x = 0; x < P; x++) for (var y = 0; y < P; y++) if (f(x, y) % P === 0) pts.push(c.point(new BN(x), new BN(y)));
  return pts;
}

function run(L, o) {
  var BN = L.curves.secp256k1.curve.p.constructor;
  if (o.op === 'tables') return render(function() {
    var c = L.curves[o.curve].curve;
    var P = tabled(L, c, o.mult, o.tamper, o.at);
    var k = new BN(o.k, 16), k2 = new BN(o.k2, 16);
    switch (o.call) {
      case 'mul': return P.mul(k);
      case 'mulAdd': return c.g.mulAdd(k, P, k2);
      case 'mulAddRev': return P.mulAdd(k, c.g, k2);
      case 'jmulAdd': return c.g.jmulAdd(k, P, k2);
      case 'derive': return new L.ec(o.curve).keyFromPrivate(o.k, 'hex').derive(P);
      case 'verify': {
        var ec = new L.ec(o.curve);
        var sg = ec.sign(o.msg, new BN(o.mult));
        return ec.verify(o.msg, sg, ec.keyFromPublic(P));
      }
      case 'eddsa-verify': {
        // (the key as a point OBJECT: eddsa/key.js:20-23 keeps it, eddsa/index.js:60-62 multiplies it)
        var ed = new L.eddsa('ed25519');
        var sig = ed.sign(o.msg, o.secret);
        var A = ed.keyFromSecret(o.secret).pub();
        var T = tabled(L, ed.curve, 1, 'null-tables', 0);
        var Q = ed.curve.point(A.getX(), A.getY());
        Q.precompute(ed.curve.n.bitLength() + 1);
        if (o.tamper === 'naf') Q.precomputed.naf.points[1 + o.at % 127] = T;
        else if (o.tamper === 'doubles') Q.precomputed.doubles.points[1 + o.at % 60] = T;
        return ed.verify(o.msg, sig, Q);
      }
      default: throw new Error('unknown call ' + o.call);
    }
  });
  if (o.op === 'g-tables') return render(function() {
    // the curve's own G with tables of its own: EC#sign / verify / recoverPubKey and EDDSA multiply IT
    var c = cloneCurve(L, o.curve);
    var G = tabled(L, c, 1, o.tamper, o.at);
    c.g = G;
    if (c.type === 'edwards') {
      var out = [];
      [ new BN(o.k, 16), new BN(o.k2, 16) ].forEach(function(k) { out.push(c.g.mul(k)); });
      out.push(c.g.mulAdd(new BN(o.k, 16), c.g.mul(new BN(5)), new BN(o.k2, 16)));
      return out;
    }
    var ec = ecOn(L, c, o.curve);
    var key = ec.keyFromPrivate(o.k, 'hex');
    var sg = ec.sign(o.msg, key);
    var good = new L.ec(o.curve).sign(o.msg, new L.ec(o.curve).keyFromPrivate(o.k, 'hex'));
    return [ sg, ec.verify(o.msg, sg, key), ec.verify(o.msg, good, new L.ec(o.curve).keyFromPrivate(o.k, 'hex').getPublic('hex'), 'hex'),
      render(function() { return ec.recoverPubKey(o.msg, good, good.recoveryParam); }), key.getPublic() ];
  });
  if (o.op === 'endo') return render(function() {
    var c = endoCurve(L, o.variant);
    var P = c.g.mul(new BN(o.mult, 16));
    var k = new BN(o.k, 16), k2 = new BN(o.k2, 16);
    return [ P.mul(k), c.g.mulAdd(k, P, k2), P.jmulAdd(k2, c.g, k), c.g.mul(k2) ];
  });
  if (o.op === 'toy-endo') return render(function() {
    // y^2 = x^3 + b over a small field with p = 1 (mod 3): the constructor finds the endomorphism by
    // itself (short.js:28-75); n is the order of g, which may be a proper divisor of the group's
    var c0 = new L.curve.short({ p: o.p.toString(16), a: '0', b: o.b.toString(16) });
    var g0 = c0.point(new BN(o.gx), new BN(o.gy)).mul(new BN(o.cof));
    var c = new L.curve.short({ p: o.p.toString(16), a: '0', b: o.b.toString(16), n: o.n.toString(16),
      g: [ g0.getX().toString(16), g0.getY().toString(16) ] });
    var out = [ c.endo ? 'endo' : 'plain' ];
    var pts = allPoints(L, c, o.p, function(x, y) { return y * y - x * x * x - o.b + 8 * o.p * o.p * o.p; });
    pts.forEach(function(q, i) {
      for (var k = 0; k <= o.p + 4; k += 1 + (i % 3)) out.push(str(q.mul(new BN(k))));
      out.push(str(c.g.mulAdd(new BN(3 + i), q, new BN(5 + 2 * i))));
    });
    return digest(out);
  });
  if (o.op === 'toy') return render(function() {
    var c, pts;
    if (o.type === 'short') {
      c = new L.curve.short({ p: o.p.toString(16), a: o.a.toString(16), b: o.b.toString(16) });
      pts = allPoints(L, c, o.p, function(x, y) { return y * y - x * x * x - o.a * x - o.b + 8 * o.p * o.p * o.p; });
    } else {
      c = new L.curve.edwards({ p: o.p.toString(16), a: o.a.toString(16), c: '1', d: o.b.toString(16) });
      pts = allPoints(L, c, o.p, function(x, y) { return o.a * x * x + y * y - 1 - o.b * x * x * y * y + 8 * o.p * o.p * o.p * o.p; });
    }
    var out = [];
    pts.forEach(function(q, i) {
      for (var k = 0; k <= 2 * o.p + 3; k++) out.push(render(function() { return q.mul(new BN(k)); }));
      out.push(render(function() { return q.mul(new BN(o.big, 16)); }));
      var r = pts[(i * 7 + 3) % pts.length];
      for (var k1 = 0; k1 < 3; k1++) for (var k2 = 0; k2 < 3; k2++)
        out.push(render(function() { return q.mulAdd(new BN(k1 + (i % 3)), r, new BN(k2 + 2 * o.p - 2)); }));
    });
    return digest(out);
  });
  if (o.op === 'private') return render(function() {
    var c = L.curves[o.curve].curve;
    var P = c.g.mul(new BN(7)), Q = c.g.mul(new BN(11));
    var k = new BN(o.k, 16), k2 = new BN(o.k2, 16);
    switch (o.call) {
      case 'fixed-no-tables': return c._fixedNafMul(P, k);
      case 'fixed-naf-only': { var T = tabled(L, c, 7, 'naf-only', 0); return c._fixedNafMul(T, k); }
      case 'wnaf-jacobian': return c._wnafMul(P.toJ(), k);
      case 'muladd-jacobian': return c._wnafMulAdd(1, [ P.toJ(), Q.toJ() ], [ k, k2 ], 2);
      case 'muladd-mixed': return c._wnafMulAdd(1, [ P, Q.toJ() ], [ k, k2 ], 2, true);
      case 'muladd-len': return c._wnafMulAdd(1, [ P, Q, c.g ], [ k, k2, k ], 2);
      case 'muladd-width': return c._wnafMulAdd(5, [ P, Q ], [ k, k2 ], 2);
      case 'endo-jacobian': return c._endoWnafMulAdd([ P.toJ() ], [ k ]);
      case 'endo-three': return c._endoWnafMulAdd([ P, Q, c.g ], [ k, k2, k ]);
      default: throw new Error('unknown call ' + o.call);
    }
  });
  if (o.op === 'foreign-red') return render(function() {
    // coordinates / signature halves that live in the reduction context of ANOTHER curve object over
    // the same field: ec.keyFromPublic(Q) builds curve.point(Q.x, Q.y) from them as they are
    // (ec/key.js:85-97), and bn.js throws 'red works only with red numbers' in the first mixed operation
    var name = o.curve, c = L.curves[name].curve, c2 = c.type === 'mont' ? null : cloneCurve(L, name);
    var k = new BN(o.k, 16);
    if (c.type === 'mont') {
      var c3 = new L.curve.mont({ p: c.p.toString(16), a: c.a.fromRed().toString(16), b: '1', g: [ '9' ] });
      var M2 = c3.g.mul(new BN(7));
      return [ render(function() { return c.point(M2.x, M2.z).mul(k); }), render(function() { return c.point(M2.getX(), new BN(1)).mul(k); }) ];
    }
    var P2 = c2.g.mul(new BN(o.mult, 16));
    if (c.type === 'edwards') {
      var ed = new L.eddsa('ed25519');
      var sig = ed.sign(o.msg, o.secret), A = ed.keyFromSecret(o.secret).pub();
      var A2 = c2.point(A.getX(), A.getY());
      return [ render(function() { return ed.verify(o.msg, sig, A2); }), render(function() { return c.point(P2.x, P2.y).mul(k); }),
        render(function() { return c.point(P2.x, P2.y, P2.z, P2.t).mul(k); }), render(function() { return c.g.mulAdd(k, c.point(P2.x, P2.y, P2.z, P2.t), k); }),
        render(function() { return c.pointFromY(P2.y, true); }), render(function() { return c.pointFromX(P2.x, false); }),
        render(function() { return c.point(P2.getX(), P2.getY()).mul(k); }) ];
    }
    var ec = new L.ec(name);
    var kp = ec.keyFromPrivate(o.mult, 'hex');
    var sg = ec.sign(o.msg, kp);
    return [ render(function() { return ec.verify(o.msg, sg, P2); }), render(function() { return ec.verify(o.msg, sg, { x: P2.x, y: P2.y }); }),
      render(function() { return ec.verify(o.msg, sg, { x: P2.getX(), y: P2.getY() }); }),
      render(function() { return ec.keyFromPrivate('0d', 'hex').derive(P2); }),
      render(function() { return ec.keyFromPublic(P2).getPublic().mul(k); }),
      render(function() { return c.g.mulAdd(k, c.point(P2.x, P2.y), k); }),
      render(function() { return c.point(P2.x, P2.y).mul(k); }),
      render(function() { return c.point(P2.x, P2.y, true).mul(k); }),
      render(function() { return ec.recoverPubKey(o.msg, { r: sg.r.toRed(c2.red), s: sg.s }, sg.recoveryParam); }),
      render(function() { return ec.recoverPubKey(o.msg, { r: sg.r, s: sg.s.toRed(c.red) }, sg.recoveryParam); }),
      render(function() { return ec.verify(o.msg, { r: sg.r.toRed(c2.red), s: sg.s }, kp); }),
      render(function() { return ec.verify(o.msg, { r: sg.r, s: sg.s.toRed(c.red) }, kp); }),
      render(function() { return c.pointFromX(P2.x, true); }), render(function() { return c.pointFromX(kp.getPublic().x, true); }),
      render(function() { return kp.getPublic().mul(k.toRed(c.red)); }), render(function() { return kp.getPublic().mul(k.toRed(c2.red)); }) ];
  });
  if (o.op === 'mutate') return render(function() { return mutateAfterUse(L, o); });
  if (o.op === 'mutate-eddsa') return render(function() {
    // an EDDSA instance's own fields, changed after its first use (eddsa/index.js:17-24)
    var ed = new L.eddsa('ed25519'), key = ed.keyFromSecret(o.secret);
    function calls() {
      return [ render(function() { var sg = ed.sign(o.msg, key); return [ sg.toHex(), ed.verify(o.msg, sg, key.getPublic()) ]; }),
        render(function() { return ed.verify(o.msg, o.sig, o.pub); }) ];
    }
    var first = calls(), undo = function() {}, second;
    function swap(obj, prop, val) { var old = obj[prop]; obj[prop] = val; undo = function() { obj[prop] = old; }; }
    switch (o.mutation) {
      case 'hash': swap(ed, 'hash', L.curves.ed25519.hash); break;                 // sha256 instead of sha512
      case 'g': swap(ed, 'g', ed.curve.g.mul(new BN(3))); break;
      case 'encodingLength': swap(ed, 'encodingLength', 31); break;
      case 'none': break;
      default: throw new Error('unknown mutation ' + o.mutation);
    }
    try { second = calls(); } finally { undo(); }
    return [ first, second, calls() ];
  });
  throw new Error('unknown op ' + o.op);
}

// ---- the same objects, changed AFTER their first use -----------------------------------------------
// The reference reads a point's tables, the endomorphism's constants, the curve's n and g and an EC
// instance's n / nh / g afresh on every call; whatever install() remembered about them after the
// first call must not outlive a change.  One recipe: the calls once (first use), the change, the
// calls again, the change undone, the calls a third time -- all three renderings are the result.
var MUTATE = [ 'entry', 'coord', 'words', 'y-words', 'width', 'beta-entry', 'beta-words', 'self-x', 'endo-basis', 'endo-beta',
  'endo-lambda', 'curve-n-words', 'curve-n-replace', 'curve-g', 'curve-b-words', 'ec-g', 'ec-n', 'ec-nh', 'none' ];
function mutateAfterUse(L, o) {
  var BN = L.curves.secp256k1.curve.p.constructor;
  // (o.preset: the library's OWN preset object -- what `new EC(name).curve` is -- instead of a copy; the
  // change is always undone before the recipe returns, whatever throws in between)
  var c = o.preset ? L.curves[o.curve].curve : o.curve === 'secp256k1' ? endoCurve(L, 'auto') : cloneCurve(L, o.curve);
  var k = new BN(o.k, 16), k2 = new BN(o.k2, 16), one = new BN(1);
  var ec = c.type !== 'short' ? null : o.preset ? new L.ec(o.curve) : ecOn(L, c, o.curve);
  if (!ec) c.g.precompute(c.n.bitLength() + 1);
  var S, other = c.g.mul(new BN(11));
  if (o.subject === 'G') S = c.g;
  else {
    var b = c.g.mul(new BN(o.mult));
    S = c.type === 'short' ? c.point(b.getX(), b.getY()) : c.point(b.getX(), b.getY());
    S.precompute(c.n.bitLength() + 1);
  }
  var wrong = c.g.mul(new BN(3));
  wrong = c.point(wrong.getX(), wrong.getY());
  var key = ec && ec.keyFromPrivate(o.d, 'hex'), good = ec && new L.ec(o.curve).sign(o.msg, o.d, 'hex');
  function calls() {
    var out = [ render(function() { return S.mul(k); }), render(function() { return S.mul(k2); }),
      render(function() { return other.mulAdd(k, S, k2); }), render(function() { return c.g.mulAdd(k2, other, k); }),
      render(function() { return c.validate(S); }) ];
    if (ec) {
      out.push(render(function() { return ec.sign(o.msg, key, { canonical: true }); }));
      out.push(render(function() { return ec.verify(o.msg, good, key.getPublic()); }));
      out.push(render(function() { return ec.verify(o.msg, good, S); }));
      out.push(render(function() { return ec.keyFromPrivate(o.d, 'hex').getPublic(); }));
      out.push(render(function() { return ec.recoverPubKey(o.msg, good, good.recoveryParam); }));
      out.push(render(function() { return key.derive(S); }));
    }
    return out;
  }
  var pre = S.precomputed, tbl = pre[o.table] || pre.naf, at = 1 + o.at % (Math.min(tbl.points.length, 120) - 1);
  var undo = function() {};
  function swap(obj, prop, val) { var old = obj[prop]; obj[prop] = val; var prev = undo; undo = function() { obj[prop] = old; prev(); }; }
  function flip(bn, word, bit) { bn.words[word] ^= bit; var prev = undo; undo = function() { bn.words[word] ^= bit; prev(); }; }
  var first = calls();
  switch (o.mutation) {
    case 'none': break;
    case 'entry': swap(tbl.points, at, wrong); break;
    case 'coord': swap(tbl.points[at], 'x', tbl.points[at].x.redAdd(c.one || one.toRed(c.red))); break;
    case 'words': flip(tbl.points[at].x, 0, 1); break;
    case 'y-words': flip(tbl.points[at].y, 1, 4); break;
    case 'width': if (tbl === pre.doubles) swap(tbl, 'step', 3); else swap(tbl, 'wnd', tbl.wnd + 1); break;
    // (precompute() leaves lambda * P WITHOUT tables -- base.js:312-327 calls _getBeta before it sets
    // this.precomputed; the G of a preset ships them, precomputed/secp256k1.js + short.js:282-310)
    case 'beta-entry':
      if (pre.beta && pre.beta.precomputed && pre.beta.precomputed.naf) swap(pre.beta.precomputed.naf.points, at, wrong);
      else if (pre.beta) swap(pre, 'beta', wrong);
      else swap(tbl.points, at, wrong);
      break;
    case 'beta-words':
      if (pre.beta && pre.beta.precomputed && pre.beta.precomputed.naf) flip(pre.beta.precomputed.naf.points[at].x, 2, 8);
      else if (pre.beta) flip(pre.beta.x, 2, 8);
      else flip(tbl.points[at].x, 2, 8);
      break;
    case 'self-x': swap(S, 'x', wrong.x); break;
    case 'endo-basis': if (c.endo) flip(c.endo.basis[o.at % 2].a, 0, 1); else flip(tbl.points[at].x, 0, 1); break;
    case 'endo-beta': if (c.endo) swap(c.endo, 'beta', c.endo.beta.redSqr()); else swap(tbl.points, at, wrong); break;
    case 'endo-lambda': if (c.endo) swap(c.endo, 'lambda', c.endo.lambda.sqr().umod(c.n)); else swap(tbl.points, at, wrong); break;
    case 'curve-n-words': flip(c.n, 0, 2); break;
    case 'curve-n-replace': swap(c, 'n', c.n.subn(2)); break;
    case 'curve-g': wrong.precompute(c.n.bitLength() + 1); swap(c, 'g', wrong); break;
    case 'curve-b-words': flip(c.type === 'short' ? c.b : c.d, 0, 1); break;
    case 'ec-g': wrong.precompute(c.n.bitLength() + 1); if (ec) swap(ec, 'g', wrong); else swap(c, 'g', wrong); break;
    case 'ec-n': if (ec) swap(ec, 'n', ec.n.subn(2)); else flip(c.n, 0, 2); break;
    case 'ec-nh': if (ec) swap(ec, 'nh', ec.nh.ushrn(3)); else flip(c.n, 0, 2); break;
    default: throw new Error('unknown mutation ' + o.mutation);
  }
  var second;
  try { second = calls(); } finally { undo(); }
  return [ first, second, calls() ];
}

// the recipes; rng: { bytes(n) -> Buffer | Array }
function recipes(rng) {
  var out = [];
  function hex(n) { return Buffer.from(rng.bytes(n)).toString('hex'); }
  function arr(n) { return Array.prototype.slice.call(rng.bytes(n)); }
  function small() { return 1 + rng.bytes(1)[0]; }
  [ 'secp256k1', 'p256', 'ed25519' ].forEach(function(curve) {
    var NB = curve === 'p256' || curve === 'secp256k1' || curve === 'ed25519' ? 32 : 0;
    TAMPER.forEach(function(tamper, ti) {
      var calls = curve === 'ed25519' ? [ 'mul', 'mulAdd', 'mulAddRev' ] : [ 'mul', 'mulAdd', 'mulAddRev', 'jmulAdd', 'derive', 'verify' ];
      calls.forEach(function(call, ci) {
        // the table a ladder reads depends on the width of k (_hasDoubles, base.js:329-338): narrow and wide
        [ hex(1 + (ti + ci) % 3), hex(NB - 1) ].forEach(function(k) {
          out.push({ op: 'tables', curve: curve, mult: 1 + (ti + ci) % 2 * 6, tamper: tamper, at: small(), call: call,
            k: k, k2: hex(NB - 1), msg: arr(32) });
        });
      });
    });
  });
  [ 'none', 'naf', 'doubles' ].forEach(function(tamper) {
    for (var i = 0; i < 3; i++)
      out.push({ op: 'tables', curve: 'ed25519', tamper: tamper, at: small(), call: 'eddsa-verify', k: '01', k2: '01', secret: hex(32), msg: arr(5 + i) });
  });
  [ 'secp256k1', 'p256', 'p384', 'ed25519' ].forEach(function(curve) {
    [ 'none', 'json', 'doubles', 'naf', 'naf-only-bad', 'beta', 'wnd-zero', 'null-tables' ].forEach(function(tamper) {
      out.push({ op: 'g-tables', curve: curve, tamper: tamper, at: small(), k: hex(20), k2: hex(31), msg: arr(32) });
    });
  });
  ENDO.forEach(function(variant) {
    for (var i = 0; i < 3; i++) out.push({ op: 'endo', variant: variant, mult: hex(8), k: hex(31), k2: hex(32 - i) });
  });
  // p = 1 (mod 3), a = 0: (p, b, a point of maximal order, cofactor taken out of it, order left)
  [ { p: 31, b: 5, gx: 1, gy: 6, cof: 3, n: 13 }, { p: 31, b: 5, gx: 1, gy: 6, cof: 1, n: 39 }, { p: 31, b: 3, gx: 1, gy: 2, cof: 1, n: 43 },
    { p: 43, b: 2, gx: 2, gy: 15, cof: 1, n: 52 }, { p: 43, b: 2, gx: 2, gy: 15, cof: 4, n: 13 }, { p: 43, b: 6, gx: 3, gy: 19, cof: 1, n: 31 } ].forEach(function(t) {
    t.op = 'toy-endo';
    out.push(t);
  });
  // every curve over F_5 and F_7 whose equation the patch must leave alone (singular cubics, Edwards
  // curves without a complete addition law) and a few it takes
  [ [ 5, 0, 0 ], [ 5, 3, 1 ], [ 5, 3, 4 ], [ 7, 0, 0 ], [ 7, 1, 2 ], [ 7, 2, 2 ], [ 7, 2, 5 ], [ 7, 4, 2 ], [ 7, 4, 5 ],
    [ 7, 1, 1 ], [ 7, 3, 3 ], [ 11, 1, 6 ], [ 13, 0, 7 ] ].forEach(function(t) {
    out.push({ op: 'toy', type: 'short', p: t[0], a: t[1], b: t[2], big: hex(32) });
  });
  [ [ 5, 2, 1 ], [ 5, 3, 4 ], [ 7, 3, 2 ], [ 7, 3, 4 ], [ 7, 5, 1 ], [ 7, 6, 4 ], [ 7, 6, 5 ], [ 7, 1, 3 ], [ 7, 2, 5 ], [ 7, 1, 2 ], [ 13, 1, 2 ], [ 13, 4, 7 ] ].forEach(function(t) {
    out.push({ op: 'toy', type: 'edwards', p: t[0], a: t[1], b: t[2], big: hex(32) });
  });
  [ 'secp256k1', 'p256' ].forEach(function(curve) {
    [ 'fixed-no-tables', 'fixed-naf-only', 'wnaf-jacobian', 'muladd-jacobian', 'muladd-mixed', 'muladd-len', 'muladd-width',
      'endo-jacobian', 'endo-three' ].forEach(function(call) {
      if (curve === 'p256' && call.slice(0, 4) === 'endo') return;
      out.push({ op: 'private', curve: curve, call: call, k: hex(31), k2: hex(16) });
    });
  });
  [ 'secp256k1', 'p256', 'p384', 'ed25519', 'curve25519' ].forEach(function(curve) {
    for (var i = 0; i < 2; i++)
      out.push({ op: 'foreign-red', curve: curve, mult: hex(12), k: hex(20 + 11 * i), msg: arr(32), secret: hex(32) });
  });
  // the same objects changed after their first use (mutateAfterUse)
  [ 'secp256k1', 'p256', 'ed25519' ].forEach(function(curve, ci) {
    MUTATE.forEach(function(mutation, mi) {
      [ 'P', 'G' ].forEach(function(subject, si) {
        if (mutation === 'self-x' && subject === 'G') return;
        out.push({ op: 'mutate', curve: curve, subject: subject, mutation: mutation, table: (ci + mi + si) % 2 ? 'naf' : 'doubles',
          at: small(), mult: 5 + (mi % 3), k: hex(31), k2: hex(1 + mi % 2 * 15), d: hex(20), msg: arr(32) });
      });
    });
  });
  [ 'hash', 'g', 'encodingLength', 'none' ].forEach(function(mutation) {
    out.push({ op: 'mutate-eddsa', mutation: mutation, secret: hex(32), msg: arr(24),
      sig: '92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00',
      pub: 'fc51cd8e6218a1a38da47ed00230f0580816ed13ba3303ac5deb911548908025' });
  });
  // ... and the library's own preset objects (`new EC(name).curve` IS elliptic.curves[name].curve)
  [ 'secp256k1', 'p256', 'ed25519' ].forEach(function(curve, ci) {
    [ 'curve-n-words', 'curve-b-words', 'endo-basis', 'endo-beta', 'curve-g', 'entry', 'words', 'none' ].forEach(function(mutation, mi) {
      out.push({ op: 'mutate', preset: true, curve: curve, subject: mi % 2 ? 'G' : 'P', mutation: mutation, table: mi % 3 ? 'naf' : 'doubles',
        at: small(), mult: 5 + (mi % 3), k: hex(31), k2: hex(1 + mi % 2 * 15), d: hex(20), msg: arr(32) });
    });
  });
  return out;
}

module.exports = { run: run, recipes: recipes, TAMPER: TAMPER, ENDO: ENDO, MUTATE: MUTATE };
