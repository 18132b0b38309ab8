'use strict';
// Loader for the reference implementation (indutny/elliptic 6.6.1): from
// $ELLIPTIC_REFERENCE, else /root/reference (build container), else the copy
// oracle/make_ref.py placed in the git-ignored oracle/_ref (that one travels to
// the GPU box).  /root/reference/lib cannot be require()d (bn.js & friends
// are not installed), but dist/elliptic.js is a browserify bundle that vendors
// them (SURVEY.md Appendix C).  We evaluate the bundle with its trailing
// "(1)" stripped so that browserify's internal require(id) is returned and
// bn.js (16), brorand (17), hash.js (19), ec/signature (10) become reachable.
//
// Used by tools/gen_golden.js, tools/run_ref_tests*.js, tools/check_patched_results.js and
// tools/bench_reference_verify.js (tests and the bench's CPU baseline).  Nothing in the
// product path touches this file.

var fs = require('fs');
var path = require('path');

var REF = path.resolve(process.env.ELLIPTIC_REFERENCE ||
  (fs.existsSync('/root/reference/dist/elliptic.js') ? '/root/reference'
    : path.join(__dirname, '..', 'oracle', '_ref')));

function load() {
  var file = path.join(REF, 'dist', 'elliptic.js');
  var src = fs.readFileSync(file, 'utf8');
  var tail = '},{},[1])(1)';
  var i = src.lastIndexOf(tail);
  if (i < 0) throw new Error('unexpected bundle layout: ' + file);
  var m = { exports: {} };
  new Function('module', 'exports', 'require',
    src.slice(0, i) + '},{},[1])' + src.slice(i + tail.length))(
    m, m.exports, require);
  var breq = m.exports;
  // The bundle ships brorand with an empty `crypto` stub; give it entropy.
  breq(17).Rand.prototype._rand = function(n) {
    return require('crypto').randomBytes(n);
  };
  return {
    root: REF,
    breq: breq,
    elliptic: breq(1),
    BN: breq(16),
    hash: breq(19),
    Signature: breq(10),
  };
}

module.exports = { load: load, REF: REF };
