'use strict';
// GPU-box self-test of the JS batch API: drives libellgpu.so through the N-API
// addon and compares with the golden fixtures (reference outputs) under
// tests/golden/.  Needs neither the reference nor mocha.
//   node elliptic_amd/js/selftest.js [libPath]
var fs = require('fs');
var path = require('path');
var ellgpu = require('./index.js');

var GOLD = path.join(__dirname, '..', '..', 'tests', 'golden');
var eng = new ellgpu.Engine({ libPath: process.argv[2] || process.env.ELLGPU_LIB });
var checked = 0;

function hexBuf(list, width) {
  return Buffer.concat(list.map(function(h) {
    var b = Buffer.from(h.length % 2 ? '0' + h : h, 'hex');
    if (b.length > width) throw new Error('value wider than field');
    return Buffer.concat([Buffer.alloc(width - b.length), b]);
  }));
}
function check(r, i, B, want, what) {
  var gotInf = r.inf[i] === 1;
  if (want.inf) { if (!gotInf) throw new Error(what + ': expected infinity at ' + i); }
  else {
    var x = r.xy.slice(i * 2 * B, i * 2 * B + B).toString('hex');
    var y = r.xy.slice(i * 2 * B + B, (i + 1) * 2 * B).toString('hex');
    if (gotInf || x !== want.x || y !== want.y) throw new Error(what + ': mismatch at ' + i);
  }
  checked++;
}

['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521', 'ed25519'].forEach(function(name) {
  var B = eng.addon.fieldBytes(eng.addon.curveId(name));
  var cases = JSON.parse(fs.readFileSync(path.join(GOLD, 'mul_' + name + '.json')));
  var fixed = cases.filter(function(c) { return c.op === 'fixed'; });
  var r = eng.mulBatch(name, hexBuf(fixed.map(function(c) { return c.k; }), B), null);
  fixed.forEach(function(c, i) { check(r, i, B, c.r, name + ' fixed'); });
  var vr = cases.filter(function(c) { return c.op === 'var'; });
  r = eng.mulBatch(name, hexBuf(vr.map(function(c) { return c.k; }), B),
    Buffer.concat(vr.map(function(c) { return hexBuf([c.px, c.py], B); })));
  vr.forEach(function(c, i) { check(r, i, B, c.r, name + ' var'); });
  var ma = cases.filter(function(c) { return c.op === 'muladd'; });
  r = eng.mulAddBatch(name, hexBuf(ma.map(function(c) { return c.k1; }), B),
    Buffer.concat(ma.map(function(c) { return hexBuf([c.p1x, c.p1y], B); })),
    hexBuf(ma.map(function(c) { return c.k2; }), B),
    Buffer.concat(ma.map(function(c) { return hexBuf([c.p2x, c.p2y], B); })));
  ma.forEach(function(c, i) { check(r, i, B, c.r, name + ' muladd'); });
  if (name === 'ed25519') return;
  var vs = JSON.parse(fs.readFileSync(path.join(GOLD, 'verify_' + name + '.json')));
  var groups = {};
  vs.forEach(function(c) { var k = c.z.length / 2; (groups[k] = groups[k] || []).push(c); });
  Object.keys(groups).forEach(function(hl) {
    var cs = groups[hl]; hl = +hl;
    var nbits = { secp256k1: 256, p192: 192, p224: 224, p256: 256, p384: 384, p521: 521 }[name];
    if (hl * 8 - Math.max(0, hl * 8 - nbits) > 32 * Math.ceil(nbits / 32)) return;
    var ok = eng.ecdsaVerifyBatch(name, { hashes: hexBuf(cs.map(function(c) { return c.z; }), hl),
      hashLen: hl, msgBits: 0, r: hexBuf(cs.map(function(c) { return c.r; }), B),
      s: hexBuf(cs.map(function(c) { return c.s; }), B),
      pub: Buffer.concat(cs.map(function(c) { return hexBuf([c.qx, c.qy], B); })) });
    cs.forEach(function(c, i) {
      if ((ok[i] === 1) !== c.ok) throw new Error(name + ' verify mismatch: ' + JSON.stringify(c));
      checked++;
    });
  });
});
// points that are not on the curve (offcurve_<curve>.json): the batch API reports them with status
// Engine.OFF_CURVE (2) and a zeroed result, and computes the on-curve items of the same batch
['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521', 'ed25519'].forEach(function(name) {
  var B = eng.addon.fieldBytes(eng.addon.curveId(name));
  var OFF = ellgpu.Engine.OFF_CURVE;
  var cases = JSON.parse(fs.readFileSync(path.join(GOLD, 'offcurve_' + name + '.json')));
  function points(r, cs, what) {
    cs.forEach(function(c, i) {
      if (c.on) return check(r, i, B, c.r, name + ' ' + what + ' (on-curve control)');
      if (r.inf[i] !== OFF) throw new Error(name + ' ' + what + ': off-curve item ' + i + ' has status ' + r.inf[i]);
      for (var b = i * 2 * B; b < (i + 1) * 2 * B; b++)
        if (r.xy[b] !== 0) throw new Error(name + ' ' + what + ': off-curve item ' + i + ' has a result');
      checked++;
    });
  }
  var vr = cases.filter(function(c) { return c.op === 'var'; });
  points(eng.mulBatch(name, hexBuf(vr.map(function(c) { return c.k; }), B),
    Buffer.concat(vr.map(function(c) { return hexBuf([c.px, c.py], B); }))), vr, 'mulBatch');
  var ma = cases.filter(function(c) { return c.op === 'muladd'; });
  points(eng.mulAddBatch(name, hexBuf(ma.map(function(c) { return c.k1; }), B),
    Buffer.concat(ma.map(function(c) { return hexBuf([c.p1x, c.p1y], B); })),
    hexBuf(ma.map(function(c) { return c.k2; }), B),
    Buffer.concat(ma.map(function(c) { return hexBuf([c.p2x, c.p2y], B); }))), ma, 'mulAddBatch');
  var vs = cases.filter(function(c) { return c.op === 'verify'; });
  if (!vs.length) return;
  var hl = vs[0].z.length / 2;
  var vst = Buffer.alloc(vs.length, 9);
  var ok = eng.ecdsaVerifyBatch(name, { hashes: hexBuf(vs.map(function(c) { return c.z; }), hl), hashLen: hl, msgBits: 0,
    r: hexBuf(vs.map(function(c) { return c.r; }), B), s: hexBuf(vs.map(function(c) { return c.s; }), B),
    pub: Buffer.concat(vs.map(function(c) { return hexBuf([c.qx, c.qy], B); })), status: vst });
  var nOff = 0;
  vs.forEach(function(c, i) {
    var inRange = !/^0*$/.test(c.r) && !/^0*$/.test(c.s) && c.note.indexOf('s = n') < 0;
    var off = !(c.on || !inRange);                              // r / s out of range win, as in the reference
    var want = off ? 0 : (c.ok ? 1 : 0);                        // the verdicts are a mask: 0 / 1, 0 for a key off the curve
    if (ok[i] !== want || vst[i] !== (off ? OFF : 0))
      throw new Error(name + ' ecdsaVerifyBatch: off-curve fixture ' + i + ' (' + c.note + ') -> ok ' + ok[i] + ' status ' + vst[i] + ', expected ' + want + ' / ' + (off ? OFF : 0));
    if (off) nOff++;
    checked++;
  });
  if (nOff < 9) throw new Error(name + ': too few off-curve verify fixtures reached the engine');
});
['secp256k1', 'p192', 'p256', 'p384', 'p521', 'ed25519'].forEach(function(name) {
  var B = eng.addon.fieldBytes(eng.addon.curveId(name));
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'decompress_' + name + '.json')));
  var r = eng.decompressBatch(name, hexBuf(cs.map(function(c) { return c.v; }), B),
    Buffer.from(cs.map(function(c) { return c.odd ? 1 : 0; })));
  cs.forEach(function(c, i) {
    var valid = !c.r.invalid;
    if ((r.ok[i] === 1) !== valid) throw new Error(name + ' decompress validity mismatch at ' + i);
    if (valid && (r.xy.slice(i * 2 * B, i * 2 * B + B).toString('hex') !== c.r.x ||
        r.xy.slice(i * 2 * B + B, (i + 1) * 2 * B).toString('hex') !== c.r.y))
      throw new Error(name + ' decompress mismatch at ' + i);
    checked++;
  });
});
['secp256k1', 'p192', 'p224', 'p256', 'p384', 'p521'].forEach(function(name) {
  var NB = eng.addon.orderBytes(eng.addon.curveId(name));
  var all = JSON.parse(fs.readFileSync(path.join(GOLD, 'sign_' + name + '.json')));
  var groups = {};
  all.forEach(function(c) { var k = (c.z.length / 2) + ':' + c.canonical; (groups[k] = groups[k] || []).push(c); });
  Object.keys(groups).forEach(function(key) {
    var cs = groups[key];
    var hl = +key.split(':')[0];
    var r = eng.ecdsaSignBatch(name, { hashes: hexBuf(cs.map(function(c) { return c.z; }), hl), hashLen: hl,
      msgBits: 0, priv: hexBuf(cs.map(function(c) { return c.d; }), NB),
      nonces: hexBuf(cs.map(function(c) { return c.k; }), NB), canonical: cs[0].canonical });
    cs.forEach(function(c, i) {
      if (c.rejected ? r.ok[i] !== 0 : (r.ok[i] !== 1 || r.r.slice(i * NB, (i + 1) * NB).toString('hex') !== c.r ||
          r.s.slice(i * NB, (i + 1) * NB).toString('hex') !== c.s || r.recid[i] !== c.recid))
        throw new Error(name + ' sign mismatch: ' + JSON.stringify(c));
      checked++;
    });
  });
});
(function() {
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'eddsa_verify_ed25519.json')));
  var r = eng.eddsaVerifyBatch(cs.map(function(c) { return Buffer.from(c.msg, 'hex'); }),
    Buffer.concat(cs.map(function(c) { return Buffer.from(c.sig, 'hex'); })),
    Buffer.concat(cs.map(function(c) { return Buffer.from(c.pub, 'hex'); })));
  cs.forEach(function(c, i) {
    if (c.throws ? !(r.err[i] === 1 && r.ok[i] === 0) : (r.err[i] !== 0 || (r.ok[i] === 1) !== c.ok))
      throw new Error('eddsa verify mismatch at ' + i);
    checked++;
  });
})();
['secp256k1', 'p256', 'p384', 'p521'].forEach(function(name) {
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'signdet_' + name + '.json')));
  var NB = cs[0].d.length / 2;
  cs.forEach(function(c) {
    var r = eng.ecdsaSignDetBatch(name, { hashes: Buffer.from(c.z, 'hex'), hashLen: c.z.length / 2,
      priv: hexBuf([c.d], NB), canonical: c.canonical });
    if (!r.ok[0] || r.r.toString('hex') !== c.r || r.s.toString('hex') !== c.s || r.recid[0] !== c.recid)
      throw new Error('deterministic sign mismatch: ' + name + ' ' + c.note);
    checked++;
  });
});
['secp256k1', 'p256', 'p384'].forEach(function(name) {
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'recover_' + name + '.json')));
  var NB = cs[0].r.length / 2;
  var byLen = {};
  cs.forEach(function(c) { (byLen[c.z.length / 2] = byLen[c.z.length / 2] || []).push(c); });
  Object.keys(byLen).forEach(function(len) {
    var g = byLen[len];
    var r = eng.ecdsaRecoverBatch(name, { hashes: hexBuf(g.map(function(c) { return c.z; }), +len), hashLen: +len,
      r: hexBuf(g.map(function(c) { return c.r; }), NB), s: hexBuf(g.map(function(c) { return c.s; }), NB),
      recid: Buffer.from(g.map(function(c) { return c.j; })) });
    var B2 = r.xy.length / g.length;
    g.forEach(function(c, i) {
      var st = r.status[i];
      var good = c.throws ? st === 2 : (c.q.inf ? st === 1 :
        (st === 0 && r.xy.slice(i * B2, (i + 1) * B2).toString('hex') === c.q.x + c.q.y));
      if (!good) throw new Error('recover mismatch: ' + name + ' ' + c.note);
      checked++;
    });
  });
});
(function() {
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'eddsa_sign_ed25519.json')));
  var r = eng.eddsaSignBatch(cs.map(function(c) { return Buffer.from(c.msg, 'hex'); }),
    Buffer.concat(cs.map(function(c) { return Buffer.from(c.secret, 'hex'); })));
  cs.forEach(function(c, i) {
    if (r.sig.slice(i * 64, i * 64 + 64).toString('hex') !== c.sig ||
        r.pub.slice(i * 32, i * 32 + 32).toString('hex') !== c.pub)
      throw new Error('eddsa sign mismatch at ' + i + ' (' + c.note + ')');
    checked++;
  });
})();
['secp256k1', 'p256', 'p521', 'ed25519'].forEach(function(name) {
  var g = JSON.parse(fs.readFileSync(path.join(GOLD, 'codec_' + name + '.json')));
  var ST = { 'Unknown point format': 1, 'invalid point': 2, 'Assertion failed': 3 };
  var byLen = {};
  g.decode.forEach(function(c) { (byLen[c.enc.length / 2] = byLen[c.enc.length / 2] || []).push(c); });
  Object.keys(byLen).forEach(function(len) {
    var cs = byLen[len];
    var r = eng.decodePointBatch(name, Buffer.from(cs.map(function(c) { return c.enc; }).join(''), 'hex'), +len);
    var B2 = r.xy.length / cs.length;
    cs.forEach(function(c, i) {
      var want = c.r.throws ? (name === 'ed25519' ? 2 : ST[c.r.throws]) : 0;
      if (r.status[i] !== want ||
          (!want && r.xy.slice(i * B2, (i + 1) * B2).toString('hex') !== c.r.x + c.r.y))
        throw new Error('decodePoint mismatch: ' + name + ' ' + c.enc);
      checked++;
    });
  });
  var xy = Buffer.from(g.encode.map(function(c) { return c.x + c.y; }).join(''), 'hex');
  var comp = eng.encodePointBatch(name, xy, true).toString('hex');
  if (comp !== g.encode.map(function(c) { return c.compact; }).join('')) throw new Error('encode(compact) mismatch: ' + name);
  if (name !== 'ed25519') {
    var full = eng.encodePointBatch(name, xy, false).toString('hex');
    if (full !== g.encode.map(function(c) { return c.full; }).join('')) throw new Error('encode mismatch: ' + name);
  }
  checked += g.encode.length;
  var B = g.encode[0].x.length / 2;
  var vs = g.validate.filter(function(c) { return c.x.length === 2 * B && c.y.length === 2 * B; });
  var st = eng.validateBatch(name, Buffer.from(vs.map(function(c) { return c.x + c.y; }).join(''), 'hex'));
  vs.forEach(function(c, i) {
    var want = name === 'ed25519' ? (c.on_curve ? (c.order_ok ? 0 : 3) : 2)
      : ellgpu.Engine.VALIDATE_REASON.indexOf(c.reason);
    if (st[i] !== want) throw new Error('validate mismatch: ' + name + ' ' + c.x);
    checked++;
  });
});
['secp256k1', 'p256', 'p521'].forEach(function(name) {
  var g = JSON.parse(fs.readFileSync(path.join(GOLD, 'wire_' + name + '.json')));
  var byLen = {};
  g.verify.forEach(function(c) {
    if (!c.key.length) return;
    (byLen[c.key.length / 2] = byLen[c.key.length / 2] || []).push(c);
  });
  Object.keys(byLen).forEach(function(len) {
    var cs = byLen[len];
    var r = eng.ecdsaVerifyWireBatch(name, { hashes: Buffer.from(cs.map(function(c) { return c.z; }).join(''), 'hex'),
      hashLen: 32, sigs: cs.map(function(c) { return Buffer.from(c.der, 'hex'); }),
      keys: Buffer.from(cs.map(function(c) { return c.key; }).join(''), 'hex'), keyLen: +len });
    cs.forEach(function(c, i) {
      var want = c.throws ? ellgpu.Engine.WIRE_ERROR.indexOf(c.throws) : 0;
      if (r.err[i] !== want || (r.ok[i] === 1) !== (c.ok === true))
        throw new Error('wire verify mismatch: ' + name + ' ' + c.note);
      checked++;
    });
  });
  var NB = g.der[0].r.length / 2;
  var ders = eng.sigToDerBatch(name, hexBuf(g.der.map(function(c) { return c.r; }), NB), hexBuf(g.der.map(function(c) { return c.s; }), NB));
  g.der.forEach(function(c, i) {
    if (ders[i].toString('hex') !== c.der) throw new Error('toDER mismatch: ' + name + ' ' + c.r);
    checked++;
  });
  var p = eng.sigFromDerBatch(name, g.parse.map(function(c) { return Buffer.from(c.der, 'hex'); }));
  g.parse.forEach(function(c, i) {
    if ((p.status[i] === 1) !== !!c.bad) throw new Error('_importDER mismatch: ' + name + ' ' + c.der);
    checked++;
  });
});
['secp256k1', 'p384', 'ed25519'].forEach(function(name) {
  var cs = JSON.parse(fs.readFileSync(path.join(GOLD, 'add_' + name + '.json')));
  var B = (cs[0].r.x || cs[0].p.x).length / 2;
  function pack(key) {
    return { xy: Buffer.concat(cs.map(function(c) { return c[key].x ? hexBuf([c[key].x, c[key].y], B) : Buffer.alloc(2 * B); })),
      inf: Buffer.from(cs.map(function(c) { return c[key].x ? 0 : 1; })) };
  }
  var p = pack('p'), q = pack('q');
  var r = eng.pointAddBatch(name, p.xy, q.xy, { infP: p.inf, infQ: q.inf });
  cs.forEach(function(c, i) {
    var good = c.r.x ? r.xy.slice(i * 2 * B, (i + 1) * 2 * B).toString('hex') === c.r.x + c.r.y &&
      (r.inf[i] === 1) === !!c.r.inf : r.inf[i] === 1;
    if (!good) throw new Error('point add mismatch: ' + name + ' ' + c.note);
    checked++;
  });
});
var lc = JSON.parse(fs.readFileSync(path.join(GOLD, 'mul_curve25519.json')));
var rr = eng.x25519Batch(hexBuf(lc.map(function(c) { return c.k; }), 32), hexBuf(lc.map(function(c) { return c.px; }), 32));
lc.forEach(function(c, i) {
  var inf = rr.inf[i] === 1;
  if (c.r.inf ? !inf : (inf || rr.x.slice(i * 32, i * 32 + 32).toString('hex') !== c.r.x))
    throw new Error('x25519 mismatch at ' + i);
  checked++;
});
// asynchronous forms must give byte-identical results, also when queued back to back
(function() {
  var B = 32;
  var cases = JSON.parse(fs.readFileSync(path.join(GOLD, 'mul_secp256k1.json')));
  var fixed = cases.filter(function(c) { return c.op === 'fixed'; });
  var vr = cases.filter(function(c) { return c.op === 'var'; });
  var kf = hexBuf(fixed.map(function(c) { return c.k; }), B);
  var kv = hexBuf(vr.map(function(c) { return c.k; }), B);
  var pv = Buffer.concat(vr.map(function(c) { return hexBuf([c.px, c.py], B); }));
  var vs = JSON.parse(fs.readFileSync(path.join(GOLD, 'verify_secp256k1.json')))
    .filter(function(c) { return c.z.length === 64; });
  var vo = { hashes: hexBuf(vs.map(function(c) { return c.z; }), 32), hashLen: 32, msgBits: 0,
    r: hexBuf(vs.map(function(c) { return c.r; }), B), s: hexBuf(vs.map(function(c) { return c.s; }), B),
    pub: Buffer.concat(vs.map(function(c) { return hexBuf([c.qx, c.qy], B); })) };
  var s1 = eng.mulBatch('secp256k1', kf, null);
  var s2 = eng.mulBatch('secp256k1', kv, pv);
  var s3 = eng.ecdsaVerifyBatch('secp256k1', vo);
  // the newer operations: deterministic sign, recovery of those signatures, wire-format verify, decodePoint
  var so = { hashes: vo.hashes, hashLen: 32, canonical: true,
    priv: hexBuf(vs.map(function(c, i) { return fixed[i % fixed.length].k; }), B) };
  var s4 = eng.ecdsaSignDetBatch('secp256k1', so);
  var ro = { hashes: vo.hashes, hashLen: 32, r: s4.r, s: s4.s, recid: s4.recid };
  var s5 = eng.ecdsaRecoverBatch('secp256k1', ro);
  var wg = JSON.parse(fs.readFileSync(path.join(GOLD, 'wire_secp256k1.json'))).verify
    .filter(function(c) { return c.key.length === 66; });
  var wo = { hashes: Buffer.from(wg.map(function(c) { return c.z; }).join(''), 'hex'), hashLen: 32,
    sigs: wg.map(function(c) { return Buffer.from(c.der, 'hex'); }),
    keys: Buffer.from(wg.map(function(c) { return c.key; }).join(''), 'hex'), keyLen: 33 };
  var s6 = eng.ecdsaVerifyWireBatch('secp256k1', wo);
  var s7 = eng.decodePointBatch('secp256k1', wo.keys, 33);
  Promise.all([eng.mulBatchAsync('secp256k1', kf, null), eng.mulBatchAsync('secp256k1', kv, pv),
    eng.ecdsaVerifyBatchAsync('secp256k1', vo),
    eng.mulBatchAsync('secp256k1', Buffer.alloc(31), null).then(function() { return 'no error'; },
      function(e) { return 'rejected: ' + e.message; }),
    eng.ecdsaSignDetBatchAsync('secp256k1', so), eng.ecdsaRecoverBatchAsync('secp256k1', ro),
    eng.ecdsaVerifyWireBatchAsync('secp256k1', wo), eng.decodePointBatchAsync('secp256k1', wo.keys, 33)])
    .then(function(r) {
      ['r', 's', 'recid', 'ok'].forEach(function(k) { if (!r[4][k].equals(s4[k])) throw new Error('async signDet differs: ' + k); });
      if (!r[5].xy.equals(s5.xy) || !r[5].status.equals(s5.status)) throw new Error('async recover differs');
      if (!r[6].ok.equals(s6.ok) || !r[6].err.equals(s6.err)) throw new Error('async wire verify differs');
      if (!r[7].xy.equals(s7.xy) || !r[7].status.equals(s7.status)) throw new Error('async decodePoint differs');
      checked += 2 * vs.length + 2 * wg.length;
      if (!r[0].xy.equals(s1.xy) || !r[0].inf.equals(s1.inf)) throw new Error('async mulFixed differs');
      if (!r[1].xy.equals(s2.xy) || !r[1].inf.equals(s2.inf)) throw new Error('async mulVar differs');
      if (!r[2].equals(s3)) throw new Error('async verify differs');
      if (r[3].indexOf('rejected') !== 0) throw new Error('bad-length async call was not rejected');
      checked += fixed.length + vr.length + vs.length;
      console.log(JSON.stringify({ ok: true, checked: checked, async: true, engine: eng.stats }));
      process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
    }).catch(function(e) { console.error(e); process.exit(1); });
})();
