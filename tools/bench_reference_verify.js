'use strict';
// The reference's own pure-JS CPU path, timed on the machine this runs on -- on the GPU box
// through the copy oracle/make_ref.py placed in oracle/_ref.  bench.py's cpu_baseline leg
// (kind "reference") starts one process of this script for the 1-core figure and then one per
// host core over disjoint slices of the SAME tuples the GPU verifies (SURVEY.md 8d row 1).
// The timed call is benchmarks/index.js:106-109's: ec.verify(msg, signature, key) with a
// ready Signature / KeyPair object.  Every verdict is compared with the expected mask, so the
// run doubles as a parity check of the bench inputs against the reference.  Checker only.
//
//   node tools/bench_reference_verify.js <tuples.bin> <first> <count> <seconds> [fixed]
// tuples.bin: records of 161 bytes = hash32 | r32 | s32 | x32 | y32 | expected(1)
var fs = require('fs');
var os = require('os');
var ref = require('./ref_loader').load();
var elliptic = ref.elliptic;

var file = process.argv[2];
var first = parseInt(process.argv[3], 10);
var count = parseInt(process.argv[4], 10);
var secs = parseFloat(process.argv[5] || '3');
var fixed = process.argv[6] === 'fixed';
var REC = 161;

var ec = new elliptic.ec('secp256k1');
var fd = fs.openSync(file, 'r');
var buf = Buffer.alloc(count * REC);
fs.readSync(fd, buf, 0, count * REC, first * REC);
fs.closeSync(fd);

function item(i) {
  var o = i * REC;
  return {
    m: buf.slice(o, o + 32),
    sig: new ref.Signature({ r: buf.slice(o + 32, o + 64).toString('hex'), s: buf.slice(o + 64, o + 96).toString('hex') }),
    key: ec.keyFromPublic({ x: buf.slice(o + 96, o + 128).toString('hex'), y: buf.slice(o + 128, o + 160).toString('hex') }),
    want: buf[o + 160] !== 0,
  };
}

// objects are built in chunks outside the timed regions (the reference's benchmark also
// verifies ready objects)
var CH = 256, done = 0, bad = 0, spent = 0;
var pos = 0;
ec.verify(item(0).m, item(0).sig, item(0).key);          // warm the JIT and G's tables
while (spent < secs && (fixed || pos < count)) {
  var n = Math.min(CH, fixed ? CH : count - pos);
  var items = [];
  for (var i = 0; i < n; i++) items.push(item(fixed ? 0 : pos + i));
  var t0 = process.hrtime.bigint();
  for (i = 0; i < n; i++) {
    if (ec.verify(items[i].m, items[i].sig, items[i].key) !== items[i].want) bad++;
  }
  spent += Number(process.hrtime.bigint() - t0) / 1e9;
  done += n;
  pos += n;
}
console.log(JSON.stringify({
  done: done, seconds: spent, per_s: done / spent, mismatches: bad, first: first,
  version: elliptic.version, node: process.version, cpu: os.cpus()[0].model, logical_cpus: os.cpus().length,
}));
