'use strict';
// eng.verifyAsync (elliptic_amd/js/index.js): N concurrent single verifications must become ONE
// engine call, with the verdicts -- and the rejections -- the reference's own synchronous
// EC#verify gives for the same arguments (ec/index.js:188-229).
//   ELLGPU_LIB=<hostsim or real library> node tools/check_verify_coalescing.js
var loader = require('./ref_loader');
var plain = loader.load().elliptic;          // unpatched reference: the judge of every verdict
var patched = loader.load().elliptic;
var eng = require('../elliptic_amd/js').install(patched, { libPath: process.env.ELLGPU_LIB });
var crypto = require('crypto');

function die(msg) { console.log(JSON.stringify({ ok: false, error: msg })); process.exit(1); }

var ecp = new plain.ec('secp256k1'), ecq = new patched.ec('secp256k1');
var ecq256 = new patched.ec('p256'), ecp256 = new plain.ec('p256');
var N = 96;
var calls = [];
for (var i = 0; i < N; i++) {
  var ec = i % 8 === 7 ? ecp256 : ecp;
  var kp = ec.genKeyPair({ entropy: crypto.createHash('sha512').update('k' + i).digest() });
  var msg = crypto.createHash(i % 5 === 4 ? 'sha1' : 'sha256').update('m' + i).digest();   // two digest lengths
  var sig = kp.sign(msg).toDER('hex');
  if (i % 6 === 5) msg = Buffer.concat([ msg.slice(0, msg.length - 1), Buffer.from([ msg[msg.length - 1] ^ 1 ]) ]);  // corrupted
  calls.push({ p256: ec === ecp256, msg: msg, sig: sig, key: kp.getPublic('hex'), enc: 'hex' });
}
// two calls the reference throws on: an undecodable public key, a malformed DER signature
calls.push({ p256: false, msg: calls[0].msg, sig: calls[0].sig, key: '05abcdef', enc: 'hex' });
// ... and a well-formed key that is not on the curve (the reference answers without validating it)
calls.push({ p256: false, msg: calls[0].msg, sig: calls[0].sig, key: '04' + '11'.repeat(64), enc: 'hex' });
calls.push({ p256: false, msg: calls[1].msg, sig: '3006020101', key: calls[1].key, enc: 'hex' });

// EC#verify's fifth argument (options.msgBitLength, ec/index.js:188-192, 97-102): a digest declared
// 260 bits long is shifted right by 4 -- its own batch (grouped by msgBitLength); an Array with a
// non-byte element and an empty message go down the reference's synchronous path (new BN(msg, 16)
// does not truncate array elements mod 256; an empty message is z = 0)
(function() {
  var kp = ecp.genKeyPair({ entropy: crypto.createHash('sha512').update('opt').digest() });
  var long = crypto.createHash('sha256').update('long').digest();
  var sig = kp.sign(long, { msgBitLength: 260 }).toDER('hex');                                       // z = H >> 4
  calls.push({ p256: false, msg: long, sig: sig, key: kp.getPublic('hex'), enc: 'hex', options: { msgBitLength: 260 } });
  calls.push({ p256: false, msg: long, sig: sig, key: kp.getPublic('hex'), enc: 'hex' });          // same bytes, no option: another z
  var odd = [ 1, 2, 300, 4 ];
  // (EC#sign asserts byte elements, EC#verify does not: any signature will do, the verdicts are compared)
  calls.push({ p256: false, msg: odd, sig: sig, key: kp.getPublic('hex'), enc: 'hex' });
  calls.push({ p256: false, msg: [], sig: kp.sign([]).toDER('hex'), key: kp.getPublic('hex'), enc: 'hex' });
})();

var want = calls.map(function(c) {
  try { return { v: (c.p256 ? ecp256 : ecp).verify(c.msg, c.sig, c.key, c.enc, c.options) }; }
  catch (e) { return { e: String(e.message) }; }
});
// warm-up: the first marshalling takes the library's Signature class from one signature, once
eng.verifyMany(ecq, [ { msg: calls[0].msg, signature: calls[0].sig, key: calls[0].key, enc: 'hex' } ]);
var before = eng.stats.gpuCalls;
var ps = calls.map(function(c) {
  return eng.verifyAsync(c.p256 ? ecq256 : ecq, c.msg, c.sig, c.key, c.enc, c.options)
    .then(function(v) { return { v: v }; }, function(e) { return { e: String(e.message) }; });
});
Promise.all(ps).then(function(got) {
  for (var i = 0; i < calls.length; i++) {
    if (JSON.stringify(got[i]) !== JSON.stringify(want[i]))
      die('call ' + i + ': ' + JSON.stringify(got[i]) + ' != reference ' + JSON.stringify(want[i]));
  }
  var launches = eng.stats.gpuCalls - before;
  // groups: (secp256k1, 32-byte), (secp256k1, 20-byte), (p256, 32-byte), (p256, 20-byte), the
  // digest with msgBitLength (its own group), and the two synchronous calls
  // (non-byte array, empty message: one patched ladder call each)
  if (launches > 8) die(calls.length + ' concurrent verifyAsync calls made ' + launches + ' engine calls');
  var nTrue = want.filter(function(w) { return w.v === true; }).length;
  var nThrow = want.filter(function(w) { return w.e !== undefined; }).length;
  if (nTrue < 60 || nThrow !== 2) die('test set degenerate: ' + nTrue + ' valid, ' + nThrow + ' throwing');
  // a second tick forms its own batch
  return eng.verifyAsync(ecq, calls[0].msg, calls[0].sig, calls[0].key, 'hex').then(function(v) {
    if (v !== want[0].v) die('second tick verdict');
    console.log(JSON.stringify({ ok: true, calls: calls.length, engine_calls: launches, valid: nTrue, rejected: nThrow,
      coalescedBatches: eng.stats.coalescedBatches, coalescedItems: eng.stats.coalescedItems }));
    process.exit(0);   // (explicit exit: node 12's environment teardown can crash in a pending N-API second-pass weak callback -- INTEGRATION.md, known issues)
  });
}).catch(function(e) { die(String(e && e.stack || e)); });
