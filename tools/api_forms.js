'use strict';
// Public-API calls whose arguments are OBJECTS of the library (points, KeyPairs, Signatures, BNs)
// rather than bytes -- the forms in which the round-4 review found the patched library answering
// differently from the reference.  A case is a RECIPE (plain JSON): run(lib, recipe) builds the
// objects inside `lib` and returns a comparable rendering of the result ('v:...' or 'e:<message>').
//   tools/gen_golden.js          runs every recipe on the unpatched reference -> tests/golden/api_forms.json
//   tools/check_patched_results.js  replays them through install() and compares
// Reference: lib/elliptic/eddsa/index.js:52-63, eddsa/key.js:20-23, eddsa/signature.js:33-38,
// ec/index.js:110-186 (sign: :133-139 the digest's width), ec/key.js:23-24,31-32.

var ED_PUB_FORMS = 13, ED_SIG_FORMS = 18;

function edBuild(L, o) {
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

function render(f) {
  try {
    var v = f();
    if (v && v.r && v.s) return 'v:sig(' + v.r.toString(16) + ',' + v.s.toString(16) + ',' + v.recoveryParam + ')';
    if (v && v.constructor && v.constructor.name === 'BN') return 'v:bn' + v.toString(16);
    return 'v:' + String(v);
  } catch (e) { return 'e:' + String(e && e.message); }
}

function run(L, o) {
  if (o.op === 'eddsa-verify') return render(function() {
    var b = edBuild(L, o);
    var first = b.ed.verify(b.msg, b.sig, b.pub);
    return o.twice ? String(first) + ',' + b.ed.verify(b.msg, b.sig, b.pub) : first;     // (the reference caches on the objects)
  });
  if (o.op === 'sign-width') return render(function() {
    var ec = new L.ec(o.curve);
    return ec.sign(o.msg, o.priv, 'hex', o.msgBitLength === null ? undefined : { msgBitLength: o.msgBitLength });
  });
  if (o.op === 'verify-width') return render(function() {
    var ec = new L.ec(o.curve);
    var opts = o.msgBitLength === null ? undefined : { msgBitLength: o.msgBitLength };
    return ec.verify(o.msg, { r: o.r, s: o.s }, ec.keyFromPrivate(o.priv, 'hex').getPublic('hex'), 'hex', opts);
  });
  var from = new L.ec(o.from), to = new L.ec(o.to);
  if (o.op === 'foreign-sign') return render(function() { return to.sign(o.msg, from.keyFromPrivate(o.priv, 'hex')); });
  if (o.op === 'foreign-keypair-sign') return render(function() { return from.keyFromPrivate(o.priv, 'hex').sign(o.msg); });
  if (o.op === 'foreign-verify') return render(function() { return to.verify(o.msg, { r: o.r, s: o.s }, from.keyFromPrivate(o.priv, 'hex')); });
  if (o.op === 'foreign-derive') return render(function() { return to.keyFromPrivate('0d', 'hex').derive(from.keyFromPrivate(o.priv, 'hex').getPublic()); });
  throw new Error('unknown op ' + o.op);
}

// the recipes; rng: { bytes(n) -> Buffer | Array }, L: the reference (for curve orders and the
// signatures some recipes carry)
function recipes(rng, L) {
  var out = [];
  function arr(n) { return Array.prototype.slice.call(rng.bytes(n)); }
  function hex(n) { return Buffer.from(rng.bytes(n)).toString('hex'); }
  // every key form x every signature form, byte messages; then the odd message forms on a sample
  for (var pf = 0; pf < ED_PUB_FORMS; pf++)
    for (var sf = 0; sf < ED_SIG_FORMS; sf++)
      out.push({ op: 'eddsa-verify', secret: hex(32), msg: arr(3 + (pf + sf) % 40), z: hex(31) + '01',
        pubForm: pf, sigForm: sf, msgForm: 0, twice: (pf + sf) % 3 === 0 });
  for (var mf = 1; mf < 4; mf++)
    [ [ 0, 0 ], [ 2, 3 ], [ 6, 11 ], [ 9, 1 ] ].forEach(function(q) {
      out.push({ op: 'eddsa-verify', secret: hex(32), msg: arr(5), z: hex(31) + '01', pubForm: q[0], sigForm: q[1], msgForm: mf, twice: false });
    });
  [ 'secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521' ].forEach(function(name) {
    var ec = new L.ec(name);
    var NB = ec.n.byteLength(), bits = ec.n.bitLength();
    var priv = hex(NB - 1);
    for (var len = NB - 1; len <= NB + 4; len++) {
      var mb = arr(len);
      mb[0] |= 0x80;
      var full = null;
      try { full = ec.sign(mb, priv, 'hex'); } catch (e) { full = null; }
      [ null, 1, 8, bits - 1, bits, bits + 1, 8 * len - 1, 8 * len, 8 * len + 8 ].forEach(function(mbl) {
        out.push({ op: 'sign-width', curve: name, msg: mb, priv: priv, msgBitLength: mbl });
        var sg = full;
        try { sg = ec.sign(mb, priv, 'hex', mbl === null ? undefined : { msgBitLength: mbl }); } catch (e) { sg = full; }
        if (sg) out.push({ op: 'verify-width', curve: name, msg: mb, priv: priv, msgBitLength: mbl, r: sg.r.toString(16), s: sg.s.toString(16) });
      });
    }
  });
  [ [ 'secp256k1', 'p256' ], [ 'p256', 'secp256k1' ], [ 'p384', 'p256' ], [ 'p224', 'p192' ], [ 'p521', 'p384' ], [ 'p256', 'p256' ] ].forEach(function(pr) {
    var from = new L.ec(pr[0]), to = new L.ec(pr[1]);
    var n2 = to.n;
    [ '07', n2.addn(5).toString(16), n2.subn(1).toString(16), n2.toString(16), hex(from.n.byteLength()) ].forEach(function(ph) {
      var msg = arr(32);
      var sg = to.sign(msg, '0b', 'hex');
      [ 'foreign-sign', 'foreign-keypair-sign', 'foreign-verify', 'foreign-derive' ].forEach(function(op) {
        out.push({ op: op, from: pr[0], to: pr[1], priv: ph, msg: msg, r: sg.r.toString(16), s: sg.s.toString(16) });
      });
    });
  });
  return out;
}

module.exports = { run: run, recipes: recipes, edBuild: edBuild };
