'use strict';
// Minimal mocha stand-in that runs the reference's own spec files
// (/root/reference/test/*.js) against the bundle loaded by ref_loader.js.
// mocha, bn.js and hash.js are not installed in this image, so the specs'
// require() calls are redirected to the bundle's internal modules
// (SURVEY.md Appendix C).  Build-container only: the GPU box has no
// /root/reference.
//
//   node tools/run_ref_tests.js            # run the suite, print a summary
//   CI=1 node tools/run_ref_tests.js       # include the long ed25519 vectors
//
// require('./run_ref_tests').run(ref, {quiet, patch}) is used by
// tools/gen_golden.js (hot-path capture) and by the install() parity gate
// (patch = function(elliptic) applied before the specs are loaded).

var path = require('path');
var Module = require('module');

function run(ref, opts) {
  opts = opts || {};
  var testDir = path.join(ref.root, 'test');
  var stats = { passed: 0, failed: 0, total: 0, failures: [] };
  var suites = [];           // stack of {name, before:[], beforeEach:[]}
  var queue = [];            // flat list of {name, fn, chain}

  function describe(name, fn) {
    suites.push({ name: name, before: [], beforeEach: [], ranBefore: false });
    fn();
    suites.pop();
  }
  function it(name, fn) {
    queue.push({ name: suites.map(function(s) { return s.name; })
      .concat([name]).join(' / '), fn: fn, chain: suites.slice() });
  }
  it.skip = function() {};
  describe.skip = function() {};
  global.describe = describe;
  global.it = it;
  global.before = function(fn) { suites[suites.length - 1].before.push(fn); };
  global.beforeEach = function(fn) {
    suites[suites.length - 1].beforeEach.push(fn);
  };
  global.after = function() {};
  global.afterEach = function() {};

  if (opts.patch) opts.patch(ref.elliptic);

  var origLoad = Module._load;
  Module._load = function(request, parent) {
    if (parent && parent.filename && parent.filename.indexOf(testDir) === 0) {
      if (request === '../') return ref.elliptic;
      if (request === 'bn.js') return ref.BN;
      if (request === 'hash.js') return ref.hash;
      if (request === '../lib/elliptic/ec/signature') return ref.Signature;
    }
    return origLoad.apply(this, arguments);
  };
  try {
    ['api-test', 'curve-test', 'ecdh-test', 'ecdsa-test', 'ed25519-test']
      .forEach(function(f) {
        var file = path.join(testDir, f + '.js');
        delete require.cache[file];
        require(file);
      });
  } finally {
    Module._load = origLoad;
  }

  var ctx = { timeout: function() {} };
  var fs = require('fs');
  // Hooks such as ed25519-test.js:48 take a `done` callback and use
  // fs.readFile; run them synchronously by serving readFile from readFileSync.
  function callHook(f) {
    if (f.length === 0) return f.call(ctx);
    var orig = fs.readFile;
    var finished = false;
    fs.readFile = function(p, cb) { cb(null, fs.readFileSync(p)); };
    try { f.call(ctx, function() { finished = true; }); } finally {
      fs.readFile = orig;
    }
    if (!finished) throw new Error('async hook did not complete synchronously');
  }
  queue.forEach(function(t) {
    stats.total++;
    try {
      t.chain.forEach(function(s) {
        if (!s.ranBefore) {
          s.ranBefore = true;
          s.before.forEach(callHook);
        }
      });
      t.chain.forEach(function(s) {
        s.beforeEach.forEach(callHook);
      });
      if (t.fn.length > 0) throw new Error('async spec not supported');
      t.fn.call(ctx);
      stats.passed++;
    } catch (e) {
      stats.failed++;
      stats.failures.push(t.name + ': ' + (e && e.message));
      if (!opts.quiet) console.log('FAIL', t.name, '\n   ', e && e.stack);
    }
  });
  return stats;
}

module.exports = { run: run };

if (require.main === module) {
  var ref = require('./ref_loader').load();
  var s = run(ref, {});
  console.log(JSON.stringify({ passed: s.passed, failed: s.failed,
    total: s.total }));
  process.exit(s.failed ? 1 : 0);
}
