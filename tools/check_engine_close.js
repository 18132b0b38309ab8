'use strict';
// Engine#close (the addon pins a context until destroyContext, INTEGRATION.md section 2): after
// uninstall() + close() the library is the reference again, the closed engine's own calls throw, a
// second close() is a no-op, a new install() gets a context of its own; an engine with a Promise-form
// batch in flight refuses to close until the Promise has settled.
//   ELLGPU_LIB=<hostsim or real library> node tools/check_engine_close.js
var loader = require('./ref_loader');
var L = loader.load().elliptic;
var js = require('../elliptic_amd/js');
function die(m) { console.log(JSON.stringify({ ok: false, error: m })); process.exit(1); }
var eng = js.install(L, { libPath: process.env.ELLGPU_LIB });
var ec = new L.ec('secp256k1'), key = ec.keyFromPrivate('5e2f7a1c9b3d48e60f1a2b3c4d5e6f708192a3b4', 'hex');
var sig = ec.sign([1, 2, 3], key);
if (!ec.verify([1, 2, 3], sig, key.getPublic()) || !eng.stats.gpuCalls) die('first engine');
var pending = eng.verifyManyAsync(ec, [ { msg: Buffer.from([1, 2, 3]), signature: sig, key: key.getPublic() } ]);
var refused = null;
try { eng.close(); } catch (e) { refused = e.message; }
pending.then(function(v) {
  if (v[0] !== true) die('async verdict');
  if (refused === null || !/in flight/.test(refused)) die('close() with a batch in flight must be refused: ' + refused);
  if (!eng.ctx) die('a refused close() must leave the engine usable');
  eng.uninstall(); eng.close(); eng.close();
  var calls = eng.stats.gpuCalls;
  if (!ec.verify([1, 2, 3], ec.sign([1, 2, 3], key), key.getPublic()) || eng.stats.gpuCalls !== calls) die('after close: not the reference');
  var threw = false;
  try { eng.mulBatch('secp256k1', Buffer.alloc(32, 1), null); } catch (e) { threw = /context/.test(e.message); }
  if (!threw) die('a closed engine must throw');
  var e2 = js.install(L, { libPath: process.env.ELLGPU_LIB });
  if (!ec.verify([1, 2, 3], sig, key.getPublic()) || !e2.stats.gpuCalls) die('second engine');
  // more engines than the addon will pin at once: each closed before the next
  for (var i = 0; i < 70; i++) { var e = new js.Engine({ libPath: process.env.ELLGPU_LIB }); e.close(); }
  console.log(JSON.stringify({ ok: true, refused_while_in_flight: true, engines_opened_and_closed: 70, engine: e2.stats }));
  process.exit(0);
}, function(e) { die(String(e && e.stack || e)); });
