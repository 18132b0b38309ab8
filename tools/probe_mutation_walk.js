'use strict';
// A WALK over everything a caller can reach from the objects of a protocol call -- the EC / EDDSA
// instance, its curve (coefficients, order, generator, endomorphism constants, reduction context,
// flags), the generator's and an operand's precomputed tables, a key pair, a signature -- that
// changes ONE reachable property AFTER the objects' first use, runs the public API on an unpatched
// copy of the reference and on a copy patched by install(), and compares every result and every
// exception message; then undoes the change and compares again.  The hand-written `mutate` recipes
// of tools/trusted_inputs.js name the changes somebody thought of; this names all of them, so that
// INTEGRATION.md's closed list ("re-read on every call" / "treated as immutable") can be checked
// against the objects instead of against memory.  A path whose change the two libraries answer
// differently must lie in the documented "treated as immutable" class (inside `curve.red`,
// `curve.type`, `_maxwellTrick` / `redN`, an EDDSA KeyPair's cached secret material); anything else is a
// parity failure.
//   ELLGPU_LIB=<hostsim or real library> node tools/probe_mutation_walk.js [family ...]
// families: short:<preset> edwards:ed25519 mont:curve25519 custom:<name of tests/golden/custom_short.json>
// customed:<name of custom_edwards.json> ec:ed25519 (the EC class over the Edwards curve) (default: short:secp256k1 short:p256 edwards:ed25519 mont:curve25519 -- the
// first, second, fourth and fifth; WALK_STRIDE=n WALK_OFFSET=k: every n-th path only).  Every path runs in a child process under a time limit (a
// change may make the REFERENCE loop -- e.g. a signing loop that never finds a nonce); a path that
// runs out of time on both libraries alike is reported as `hang` and skipped.
var cp = require('child_process');
var fs = require('fs');
var crypto = require('crypto');

// INTEGRATION.md section 2, "treated as immutable": the reduction context, curve.type, _maxwellTrick / redN,
// and what an EDDSA KeyPair caches of its secret (eddsa/key.js:39-71: the engine signs from secret())
var IMMUTABLE_SEGMENTS = { red: 1, type: 1, _maxwellTrick: 1, redN: 1 };
var KEYPAIR_CACHES = { _secret: 1, _priv: 1, _pubBytes: 1, _messagePrefix: 1, _hash: 1 };
function documentedImmutable(path, family) {
  if (family.split(':')[0] === 'edwards' && path[0] === 'key' && KEYPAIR_CACHES[path[1]] === 1) return true;
  return path.some(function(s) { return IMMUTABLE_SEGMENTS[s] === 1; });
}

function kindOf(v) {
  if (v === null || v === undefined) return 'nil';
  var t = typeof v;
  if (t === 'boolean' || t === 'number' || t === 'string' || t === 'function') return t;
  if (Array.isArray(v)) return 'array';
  if (v.constructor && v.constructor.name === 'BN') return 'bn';
  if (typeof v.isInfinity === 'function' && v.curve) return 'point';
  if (ArrayBuffer.isView(v)) return 'bytes';
  return 'object';
}
function sample(n) {                                    // indices of an array worth visiting
  var s = {}; [ 0, 1, 2, n >> 1, n - 2, n - 1 ].forEach(function(i) { if (i >= 0 && i < n) s[i] = 1; });
  return Object.keys(s).map(Number);
}
function walk(roots, maxDepth) {
  var seen = new Set(), out = [];
  function visit(v, path) {
    var k = kindOf(v);
    if (k === 'nil' || k === 'string' || k === 'function' || k === 'bytes') return;
    if (k === 'boolean') { out.push({ path: path, form: 'flip' }); out.push({ path: path, form: 'delete' }); return; }
    if (k === 'number') { out.push({ path: path, form: 'inc' }); out.push({ path: path, form: 'delete' }); return; }
    if (seen.has(v)) return;
    seen.add(v);
    if (path.length > 1) {
      if (k === 'bn' || k === 'point') out.push({ path: path, form: 'replace' });
      else out.push({ path: path, form: 'null' });
      out.push({ path: path, form: 'delete' });
    }
    if (k === 'bn') {
      out.push({ path: path, form: 'w0' });
      out.push({ path: path, form: 'neg' });
      if (v.length > 1) out.push({ path: path, form: 'trunc' });
      // WALK_PAD=1: the same value with a leading zero word.  bn.js keeps its numbers stripped and its own methods
      // (cmpn, dist/elliptic.js:6675) strip in place whatever they are asked about: a number in that state is changed by
      // whoever looks at it -- INTEGRATION.md section 2 lists broken bn.js invariants with the swapped `words` array
      if (process.env.WALK_PAD === '1') out.push({ path: path, form: 'pad' });
      return;
    }
    if (path.length > maxDepth) return;
    if (k === 'array') {
      if (v.length > 1) out.push({ path: path, form: 'pop' });
      sample(v.length).forEach(function(i) { visit(v[i], path.concat([ i ])); });
      return;
    }
    Object.keys(v).forEach(function(key) { visit(v[key], path.concat([ key ])); });
  }
  Object.keys(roots).forEach(function(r) { visit(roots[r], [ r ]); });
  return out;
}
function resolve(roots, path) { var v = roots; path.forEach(function(s) { v = v[s]; }); return v; }

// ---- rendering (results and exception messages as comparable strings) -----------------------------
function pointStr(v) {
  if (v.isInfinity() && v.curve.type !== 'edwards') return 'O';
  if (v.curve.type === 'mont') return 'x=' + v.getX().toString(16);
  if (v.curve.type === 'edwards') { var q = v.curve.point(v.x, v.y, v.z, v.t); return '(' + q.getX().toString(16) + ',' + q.getY().toString(16) + ')'; }
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
function render(f) { try { return 'v:' + str(f()); } catch (e) { return 'e:' + String(e && e.message).slice(0, 120); } }

// ---- the objects and the calls of a family --------------------------------------------------------
var K = 'a3f1c29b7d5e08416c2a9f13e57b8d60412f9a7c3e5b1d08f6a2c4e19b3d5f71', K2 = '1d', D = '5e2f7a1c9b3d48e60f1a2b3c4d5e6f708192a3b4c5d6e7f8091a2b3c4d5e6f70';
var D2 = '0b1c2d3e4f5061728394a5b6c7d8e9fa0b1c2d3e4f5061728394a5b6c7d8e9f1';
var MSG = []; for (var mi = 0; mi < 32; mi++) MSG.push((mi * 37 + 11) & 255);

function setup(L, family, eng) {
  var kind = family.split(':')[0], name = family.split(':')[1];
  var BN = L.curves.secp256k1.curve.p.constructor;
  var k = new BN(K, 16), k2 = new BN(K2, 16);
  if (kind === 'custom' || kind === 'customed') {
    // a USER-DEFINED curve (tests/golden/custom_short.json / custom_edwards.json: run-time modulus on the
    // device, Point#mul / mulAdd / jmulAdd only -- its protocol calls are the reference's own)
    var path = require('path');
    var specs = JSON.parse(fs.readFileSync(path.join(__dirname, '..', 'tests', 'golden', kind === 'custom' ? 'custom_short.json' : 'custom_edwards.json')));
    var sp = specs.filter(function(x) { return x.name === name; })[0];
    var cc = kind === 'custom' ? new L.curve.short({ p: sp.p, a: sp.a, b: sp.b, n: sp.n, g: [ sp.g.x, sp.g.y ] }) :
      new L.curve.edwards({ p: sp.p, a: sp.a, c: '1', d: sp.d, n: sp.n || null, g: [ sp.g.x, sp.g.y ] });
    var nb = cc.n ? cc.n.bitLength() : cc.p.bitLength();
    var bc = cc.g.mul(new BN(7)), Sc = cc.point(bc.getX(), bc.getY());
    Sc.precompute(nb + 1);
    cc.g.precompute(nb + 1);
    var oc = cc.g.mul(new BN(11)), otherc = cc.point(oc.getX(), oc.getY());
    var wc = cc.g.mul(new BN(3)), wrongc = cc.point(wc.getX(), wc.getY());
    var kc = k.ushrn(256 - Math.min(nb, 256) + 2);
    var ecc = kind === 'custom' && cc.n ? new L.ec({ curve: { curve: cc, g: cc.g, n: cc.n, hash: L.curves.p256.hash } }) : null;
    var keyc = ecc && ecc.keyFromPrivate(kc.toString(16), 'hex'), goodc = ecc && ecc.sign(MSG, keyc, { canonical: true });
    var rootsC = { curve: cc, S: Sc, other: otherc };
    if (ecc) { rootsC = { ec: ecc, S: Sc, other: otherc, key: keyc, good: goodc }; }
    return { roots: rootsC, wrong: wrongc, BN: BN, calls: function() {
      var out = [ render(function() { return Sc.mul(kc); }), render(function() { return Sc.mul(k2); }), render(function() { return cc.g.mul(kc); }),
        render(function() { return otherc.mulAdd(kc, Sc, k2); }), render(function() { return cc.g.mulAdd(k2, otherc, kc); }),
        render(function() { return cc.validate(Sc); }), render(function() { return Sc.add(otherc); }) ];
      if (typeof otherc.jmulAdd === 'function') out.push(render(function() { return otherc.jmulAdd(kc, Sc, k2); }));
      if (ecc) {
        out.push(render(function() { return ecc.sign(MSG, keyc, { canonical: true }); }));
        out.push(render(function() { return ecc.verify(MSG, goodc, keyc.getPublic()); }));
        out.push(render(function() { return ecc.keyFromPrivate(kc.toString(16), 'hex').getPublic(); }));
        out.push(render(function() { return ecc.recoverPubKey(MSG, goodc, goodc.recoveryParam); }));
        out.push(render(function() { return keyc.derive(Sc); }));
      }
      return out;
    } };
  }
  if (kind === 'ec') {
    // the EC class over a curve of ANOTHER model (`new EC('ed25519')`: ECDSA over the Edwards curve, which
    // the reference supports -- test/ecdsa-test.js -- and whose protocol calls are its own; the ladders
    // under them are the engine's)
    var e2 = new L.ec(name), c2 = e2.curve;
    var key2 = e2.keyFromPrivate(D.slice(0, 40), 'hex'), good2 = e2.sign(MSG, key2, { canonical: true });
    var other2 = e2.keyFromPrivate(D2.slice(0, 40), 'hex').getPublic();
    var w2 = c2.g.mul(new BN(3));
    return { roots: { ec: e2, key: key2, good: good2, other: other2 }, wrong: w2, BN: BN, calls: function() {
      return [ render(function() { return e2.sign(MSG, key2, { canonical: true }); }),
        render(function() { return e2.verify(MSG, good2, key2.getPublic()); }),
        render(function() { return e2.verify(MSG, good2, other2); }),
        render(function() { return e2.keyFromPrivate(D.slice(0, 40), 'hex').getPublic(); }),
        render(function() { return key2.derive(other2); }),
        render(function() { return c2.g.mul(k); }), render(function() { return other2.mul(k2); }),
        render(function() { return c2.validate(other2); }) ];
    } };
  }
  if (kind === 'short') {
    var ec = new L.ec(name), c = ec.curve;
    var b = c.g.mul(new BN(7)), S = c.point(b.getX(), b.getY());
    S.precompute(c.n.bitLength() + 1);
    var o = c.g.mul(new BN(11)), other = c.point(o.getX(), o.getY());
    var w = c.g.mul(new BN(3)), wrong = c.point(w.getX(), w.getY());
    var key = ec.keyFromPrivate(D.slice(0, 40), 'hex'), good = ec.sign(MSG, key, { canonical: true });
    var pubHex = key.getPublic(true, 'hex'), xs = S.getX();
    return { roots: { ec: ec, S: S, other: other, key: key, good: good }, wrong: wrong, BN: BN, calls: function() {
      return [ render(function() { return S.mul(k); }), render(function() { return S.mul(k2); }), render(function() { return c.g.mul(k); }),
        render(function() { return other.mulAdd(k, S, k2); }), render(function() { return c.g.mulAdd(k2, other, k); }),
        render(function() { return c.validate(S); }),
        render(function() { return ec.sign(MSG, key, { canonical: true }); }),
        render(function() { return ec.verify(MSG, good, key.getPublic()); }),
        render(function() { return ec.verify(MSG, good, S); }),
        render(function() { return ec.verify(MSG, good.toDER('hex'), pubHex, 'hex'); }),
        render(function() { return ec.keyFromPrivate(D.slice(0, 40), 'hex').getPublic(); }),
        render(function() { return ec.recoverPubKey(MSG, good, good.recoveryParam); }),
        render(function() { return ec.getKeyRecoveryParam(MSG, good, key.getPublic()); }),
        render(function() { return key.derive(S); }),
        render(function() { return c.pointFromX(xs, true); }),
        render(function() { return c.decodePoint(pubHex, 'hex'); }),
        // the patch's own batch call against what it stands for: EC#verify of every item
        render(function() {
          var items = [ { msg: MSG, signature: good, key: key.getPublic() }, { msg: MSG, signature: good, key: S },
            { msg: MSG, signature: good.toDER('hex'), key: pubHex, enc: 'hex' } ];
          return eng ? eng.verifyMany(ec, items) : items.map(function(it) { return ec.verify(it.msg, it.signature, it.key, it.enc); });
        }) ];
    }, later: function() {
      // ... and its coalescing Promise form: three calls in one tick against three EC#verify
      var items;
      try { items = [ [ MSG, good, key.getPublic() ], [ MSG, good, S ], [ MSG, good.toDER('hex'), pubHex, 'hex' ] ]; }
      catch (e) { return Promise.resolve('e:' + String(e && e.message).slice(0, 120)); }
      if (!eng) return Promise.resolve(items.map(function(it) { return render(function() { return ec.verify(it[0], it[1], it[2], it[3]); }); }).join(';'));
      return Promise.all(items.map(function(it) {
        var pr;
        try { pr = eng.verifyAsync(ec, it[0], it[1], it[2], it[3]); } catch (e) { return 'e:' + String(e && e.message).slice(0, 120); }
        return pr.then(function(v) { return 'v:' + str(v); }, function(e) { return 'e:' + String(e && e.message).slice(0, 120); });
      })).then(function(r) { return r.join(';'); });
    } };
  }
  if (kind === 'edwards') {
    var ed = new L.eddsa(name), ce = ed.curve;
    var be = ce.g.mul(new BN(7)), Se = ce.point(be.getX(), be.getY());
    Se.precompute(ce.n.bitLength() + 1);
    var oe = ce.g.mul(new BN(11)), othere = ce.point(oe.getX(), oe.getY());
    var we = ce.g.mul(new BN(3)), wronge = ce.point(we.getX(), we.getY());
    var keye = ed.keyFromSecret(D), sige = keye.sign(MSG), sigHex = sige.toHex(), pube = keye.getPublic('hex');
    var ys = Se.getY(), enc = ed.encodePoint(Se);
    // (a Signature's R / a KeyPair's point, private scalar and prefix are decoded or derived at their
    // first use and kept -- eddsa/signature.js:33-38, eddsa/key.js:39-71 --; the reference's verify / sign fill
    // them, a call the engine answers has no need to: made here, so that both libraries start from
    // objects in the same state.  INTEGRATION.md section 2 says so.)
    sige.R(); sige.S(); keye.pub(); keye.priv(); keye.messagePrefix(); keye.pubBytes();
    return { roots: { eddsa: ed, S: Se, other: othere, key: keye, sig: sige }, wrong: wronge, BN: BN, calls: function() {
      return [ render(function() { return Se.mul(k); }), render(function() { return Se.mul(k2); }), render(function() { return ce.g.mul(k); }),
        render(function() { return othere.mulAdd(k, Se, k2); }), render(function() { return ce.validate(Se); }),
        render(function() { return ed.sign(MSG, keye); }), render(function() { return ed.sign(MSG, D); }),
        render(function() { return ed.verify(MSG, sige, keye); }), render(function() { return ed.verify(MSG, sigHex, pube); }),
        render(function() { return ed.keyFromSecret(D).getPublic('hex'); }),
        render(function() { return ce.pointFromY(ys, Se.getX().isOdd()); }),
        render(function() { return ed.decodePoint(enc); }), render(function() { return ed.encodePoint(Se).join(','); }) ];
    } };
  }
  if (kind === 'mont') {
    var em = new L.ec(name), cm = em.curve;
    var keym = em.keyFromPrivate(D, 'hex'), Sm = em.keyFromPrivate(D2, 'hex').getPublic();
    var wm = cm.g.mul(new BN(3));
    return { roots: { ec: em, S: Sm, key: keym }, wrong: wm, BN: BN, calls: function() {
      return [ render(function() { return Sm.mul(k); }), render(function() { return cm.g.mul(k); }),
        render(function() { return keym.derive(Sm); }), render(function() { return em.keyFromPrivate(D, 'hex').getPublic(); }),
        render(function() { return cm.validate(Sm); }), render(function() { return Sm.getX(); }),
        render(function() { return em.keyFromPrivate(D2, 'hex').derive(keym.getPublic()); }) ];
    } };
  }
  throw new Error('unknown family ' + family);
}

function apply(ctx, item) {                               // -> undo()
  var parent = resolve(ctx.roots, item.path.slice(0, -1)), prop = item.path[item.path.length - 1], v = parent[prop];
  switch (item.form) {
    case 'flip': parent[prop] = !v; return function() { parent[prop] = v; };
    case 'inc': parent[prop] = v + 1; return function() { parent[prop] = v; };
    case 'null': parent[prop] = null; return function() { parent[prop] = v; };
    case 'delete': {
      if (Array.isArray(parent)) { delete parent[prop]; return function() { parent[prop] = v; }; }
      // (back in its place among the keys: the walk addresses array elements by index, objects by name)
      delete parent[prop]; return function() { parent[prop] = v; };
    }
    case 'pad': { var l0 = v.length; v.words[l0] = 0; v.length = l0 + 1; return function() { v.length = l0; }; }
    case 'pop': { var e = v[v.length - 1]; v.length--; return function() { v.push(e); }; }
    case 'w0': v.words[0] ^= 1; return function() { v.words[0] ^= 1; };
    case 'neg': v.negative ^= 1; return function() { v.negative ^= 1; };
    case 'trunc': { var l = v.length; v.length = l - 1; return function() { v.length = l; }; }
    case 'replace':
      if (kindOf(v) === 'bn') parent[prop] = v.red ? v.redAdd(new ctx.BN(1).toRed(v.red)) : v.addn(1);
      else parent[prop] = ctx.wrong;
      return function() { parent[prop] = v; };
  }
  throw new Error('unknown form ' + item.form);
}

// ---- child: paths [from, to) of one family ----------------------------------------------------------
async function child(family, from, to, listOnly) {
  async function all(ctx) { var r = ctx.calls(); if (ctx.later) r.push(await ctx.later()); return r; }
  var loader = require('./ref_loader');
  var A = loader.load(), B = loader.load();
  var eng = require('../elliptic_amd/js').install(B.elliptic, { libPath: process.env.ELLGPU_LIB });
  var a = setup(A.elliptic, family, null), b = setup(B.elliptic, family, eng);
  // (WALK_FRESH=1: the change comes BEFORE the operand's and the instance's first use in the calls
  // below -- no remembered verdict to go stale, the checks themselves are what is exercised; setup has
  // used the curve and G)
  var fresh = process.env.WALK_FRESH === '1';
  var first = await all(a), firstB = await all(b);
  if (fresh && !listOnly) { a = setup(A.elliptic, family, null); b = setup(B.elliptic, family, eng); }
  var items = walk(a.roots, 8);
  var stride = Number(process.env.WALK_STRIDE || 1), offset = Number(process.env.WALK_OFFSET || 0);
  if (stride > 1) items = items.filter(function(_, i) { return i % stride === offset % stride; });
  function say(o) { fs.writeSync(1, JSON.stringify(o) + '\n'); }
  if (process.env.WALK_PRINT) items.forEach(function(it, i) { say({ i: i, path: it.path.join('.'), form: it.form }); });
  if (listOnly) { say({ count: items.length, firstSame: first.join('|') === firstB.join('|'), calls: first.length }); eng.close && eng.close(); return; }
  for (var i = from; i < Math.min(to, items.length); i++) {
    var it = items[i];
    say({ start: i, path: it.path.join('.'), form: it.form });
    var ua, ub, ra, rb, na, nb;
    try { ua = apply(a, it); ub = apply(b, it); } catch (e) { say({ done: i, skipped: String(e.message) }); if (ua) ua(); continue; }
    if (process.env.WALK_ONLY === 'plain') { ra = await all(a); ua(); ub(); await all(a); say({ done: i, plainOnly: true }); continue; }
    ra = await all(a); rb = await all(b);
    ua(); ub();
    na = await all(a); nb = await all(b);
    var diff = [], stale = [];
    for (var j = 0; j < ra.length; j++) if (ra[j] !== rb[j]) diff.push({ call: j, reference: ra[j], patched: rb[j] });
    // (after the undo: the two libraries must agree again; where BOTH moved away from their first
    // answers alike -- the reference normalises points in place, caches lazily -- that is `drift`)
    var drift = 0;
    for (j = 0; j < na.length; j++) {
      if (na[j] !== nb[j]) stale.push({ call: j, first: first[j], reference: na[j], patched: nb[j] });
      else if (na[j] !== first[j]) drift++;
    }
    var changed = 0;
    for (j = 0; j < ra.length; j++) if (ra[j] !== first[j]) changed++;
    say({ done: i, path: it.path.join('.'), form: it.form, changed: changed, drift: drift, diff: diff, stale: stale });
    if (fresh) { say({ restart: i + 1 }); break; }
    if (diff.length || stale.length) { say({ restart: i + 1 }); break; }   // after a difference the next path starts from fresh objects
    first = na;
  }
  eng.close && eng.close();
}

// ---- parent ---------------------------------------------------------------------------------------------
function runChild(args, limitMs, env) {
  return new Promise(function(resolveP) {
    var p = cp.spawn(process.execPath, [ '--max-old-space-size=1024', __filename, '--child' ].concat(args),
      { stdio: [ 'ignore', 'pipe', 'inherit' ], env: Object.assign({}, process.env, env || {}) });
    var buf = '', lines = [], timer = null, killed = false;
    function arm() { if (timer) clearTimeout(timer); timer = setTimeout(function() { killed = true; p.kill('SIGKILL'); }, limitMs); }
    arm();
    p.stdout.on('data', function(d) {
      buf += d;
      var at;
      while ((at = buf.indexOf('\n')) >= 0) {
        // (a child that runs out of heap -- a change can make the reference allocate without end -- dies with
        // half a line, or V8's own text, on its stdout: not ours to parse; its exit code tells)
        var ln = null;
        try { ln = JSON.parse(buf.slice(0, at)); } catch (e) { ln = null; }
        if (ln && typeof ln === 'object') lines.push(ln);
        buf = buf.slice(at + 1); arm();
      }
    });
    p.on('close', function(code) { clearTimeout(timer); resolveP({ lines: lines, killed: killed, code: code }); });
  });
}
async function parent(families) {
  var limit = Number(process.env.WALK_LIMIT_MS || 60000), chunk = Number(process.env.WALK_CHUNK || 60);
  var summary = [], bad = 0;
  for (var fi = 0; fi < families.length; fi++) {
    var family = families[fi];
    var head = await runChild([ family, '0', '0', 'list' ], 300000);
    var info = head.lines[0];
    if (!info || !info.firstSame) { console.log(JSON.stringify({ family: family, error: 'first calls differ or the child failed', info: info })); bad++; continue; }
    var at = 0, same = 0, changedSame = 0, immutable = [], failures = [], hangs = [], skipped = 0, skippedWhy = {};
    while (at < info.count) {
      var r = await runChild([ family, String(at), String(Math.min(at + chunk, info.count)) ], limit);
      var lastStart = null, doneUpTo = at;
      var restartAt = -1;
      var take = function(l) {
        if (l.skipped) { skipped++; skippedWhy[l.skipped] = (skippedWhy[l.skipped] || 0) + 1; return; }
        var path = l.path.split('.');
        if (!l.diff.length && !l.stale.length) { same++; if (l.changed) changedSame++; return; }
        var rec = { path: l.path, form: l.form, diff: l.diff.slice(0, 2), stale: l.stale.slice(0, 2), ndiff: l.diff.length, nstale: l.stale.length };
        if (documentedImmutable(path, family)) immutable.push(rec); else failures.push(rec);
      };
      r.lines.forEach(function(l) {
        if (l.start !== undefined) { lastStart = l; return; }
        if (l.restart !== undefined) { restartAt = l.restart; return; }
        doneUpTo = l.done + 1;
        take(l);
      });
      if (r.killed || r.code !== 0) {
        if (lastStart && lastStart.start >= doneUpTo) {
          // whose hang?  the same path on the unpatched library alone
          var alone = await runChild([ family, String(lastStart.start), String(lastStart.start + 1) ], limit, { WALK_ONLY: 'plain' });
          var rec2 = { path: lastStart.path, form: lastStart.form, exit: r.killed ? 'time limit' : r.code,
            reference_alone: alone.killed ? 'time limit' : alone.code !== 0 ? 'exit ' + alone.code : 'returns' };
          if (rec2.reference_alone === 'returns') {
            // (a slow moment of the machine, or the patched library's hang: once more, by itself, with four times the limit)
            var again = await runChild([ family, String(lastStart.start), String(lastStart.start + 1) ], 4 * limit);
            var line = again.lines.filter(function(l) { return l.done === lastStart.start; })[0];
            if (line && !again.killed) take(line);
            else failures.push({ path: rec2.path, form: rec2.form, diff: [], stale: [], ndiff: 0, nstale: 0, hang: 'the patched library does not return, the reference does' });
          } else hangs.push(rec2);
          at = lastStart.start + 1;
        } else {
          // (the child ended between two paths, or before its first: nothing is skipped silently)
          if (doneUpTo === at) hangs.push({ path: '(path ' + at + ': the child ended before it started)', form: '-', exit: r.killed ? 'time limit' : r.code, reference_alone: '-' });
          at = doneUpTo + (doneUpTo === at ? 1 : 0);
        }
      } else if (restartAt >= 0) at = restartAt;
      else at = Math.min(at + chunk, info.count);
    }
    var row = { family: family, paths: info.count, calls_per_path: info.calls, same: same, same_and_the_reference_answer_changed: changedSame,
      differ_in_documented_immutable_class: immutable.length, failures: failures.length, hangs_or_crashes: hangs.length, skipped: skipped };
    console.log(JSON.stringify(row));
    Object.keys(skippedWhy).forEach(function(w) { console.log('  skipped (the change itself throws) x' + skippedWhy[w] + ': ' + w); });
    immutable.forEach(function(x) { console.log('  immutable-class ' + x.path + ' [' + x.form + '] ' + x.ndiff + ' calls differ'); });
    hangs.forEach(function(x) { console.log('  hang/crash ' + x.path + ' [' + x.form + '] ' + x.exit + '; the reference alone: ' + x.reference_alone); });
    failures.forEach(function(x) { console.log('  FAILURE ' + JSON.stringify(x)); });
    bad += failures.length;
    summary.push(row);
  }
  var dg = crypto.createHash('sha256').update(JSON.stringify(summary)).digest('hex').slice(0, 16);
  console.log(JSON.stringify({ families: summary.length, failures: bad, digest: dg }));
  process.exit(bad ? 1 : 0);
}

if (process.argv[2] === '--child') child(process.argv[3], Number(process.argv[4]), Number(process.argv[5]), process.argv[6] === 'list').then(function() { process.exit(0); },   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
  function(e) { console.error(e); process.exit(3); });
else parent(process.argv.slice(2).length ? process.argv.slice(2) : [ 'short:secp256k1', 'short:p256', 'edwards:ed25519', 'mont:curve25519' ]);
