'use strict';
// Parity gate for the JS host layer: run the reference's OWN mocha suite
// (unmodified spec files) against the reference library with
// elliptic_amd/js install() applied.  In this GPU-less container the N-API
// addon is pointed at the CPU unit-test build of the device code
// (tests/hostsim); on a box with an MI355X and the reference checked out, drop
// ELLGPU_LIB to run the same gate on the real kernels.
//
//   ELLGPU_LIB=tests/hostsim/_build/libellgpu_hostsim.so node tools/run_ref_tests_patched.js
var ref = require('./ref_loader').load();
var eng;
var s = require('./run_ref_tests').run(ref, {
  patch: function(elliptic) {
    eng = require('../elliptic_amd/js').install(elliptic, { libPath: process.env.ELLGPU_LIB });
  },
});
console.log(JSON.stringify({ passed: s.passed, failed: s.failed, total: s.total, engine: eng.stats }));
process.exit(s.failed ? 1 : 0);
