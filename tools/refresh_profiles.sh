#!/bin/bash
# Runs on the GPU box (via gpurun): every measurement DESIGN.md section 6 cites, written under
# gpurun_out/refresh/.  `python tools/refresh_profiles.py --round rNN` then copies the summaries
# into profiles/ with the round prefix and distils profiles/rNN_kernel_counters.json (what
# bench.py's roofline block reads).  PMC counters are collected in their own passes
# (--kernel-trace only; FETCH_SIZE and WRITE_SIZE do not fit one pass).
#   bash tools/refresh_profiles.sh [quick]
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/refresh
MODE="$1"
rm -rf $O && mkdir -p $O
ROUND="${ELL_ROUND:-r06}"
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
# the workload of the PMC passes: one pass of every benchmarked kernel (headline + configs), 2 timed steps
PROF="python bench.py --steps 2 --warmup 1 --no-cpu --sustain 0"
# kernel durations: the kernels by themselves (--in-flight 1: what roofline.kernel_ms is) and as the
# default command runs them (two passes in flight: spans of overlapping kernels, roofline.timed_region)
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python bench.py --steps 30 --warmup 5 --no-cpu --no-configs --in-flight 1 --sustain 0 > $O/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_stats2 -o stats2 -- python bench.py --steps 30 --warmup 5 --no-cpu --no-configs --sustain 0 > $O/bench_under_rocprof_two_in_flight.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $O/prof_sqa -o sqa -- $PROF > $O/pmc_sqa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_ANY -d $O/prof_sqb -o sqb -- $PROF > $O/pmc_sqb.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fwa -o fwa -- $PROF > $O/pmc_fwa.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_fwb -o fwb -- $PROF > $O/pmc_fwb.log 2>&1
# gather calibration of FETCH_SIZE (known bytes, this kernel's access pattern: 64 B per lane from a
# table larger than the Infinity Cache)
if [ -x tools/microbench/_build/gather_calib ]; then
  timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_gc -o gc -- tools/microbench/_build/gather_calib > $O/gather_calib.log 2>&1
fi
for t in stats stats2 sqa sqb fwa fwb gc; do
  db=$(find $O/prof_$t -name "*_results.db" 2>/dev/null | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" > $O/rocprof_$t.txt 2>&1
done
cat $O/rocprof_fwa.txt $O/rocprof_fwb.txt > $O/rocprof_fw.txt 2>/dev/null
rm -rf $O/prof_stats $O/prof_stats2 $O/prof_fwa $O/prof_fwb $O/prof_sqa $O/prof_sqb $O/prof_gc
# distil the counters HERE first (into this box's copy of profiles/), so that a bench.py run
# without its own live passes prices its roofline with the instruction counts of these binaries
python tools/refresh_profiles.py --round $ROUND --src $O > $O/distil.log 2>&1
# the smoke gate, in place and from a copy of the tree at another path (provenance must not
# depend on where the tree lives)
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
rm -rf /tmp/treecopy && mkdir -p /tmp/treecopy && cp -rL . /tmp/treecopy/repo2 2>/dev/null
( cd /tmp/treecopy/repo2 && echo "# same tree copied to $PWD" && python -c "import __graft_entry__ as g; g.smoke()" ) >> $O/smoke.log 2>&1; echo "rc=$?" >> $O/smoke.log
rm -rf /tmp/treecopy
# the driver's command: default bench.py (live PMC passes of its own, configs, cpu_baseline)
( time timeout 900 python bench.py --keep-counters $O/live_counters ) > $O/bench.json.log 2> $O/bench.err
tail -c 600 $O/bench.json.log
timeout 300 python tools/strong_proxy.py > $O/strong_proxy.jsonl 2> $O/strong_proxy.err
timeout 300 python tools/two_stream_probe.py > $O/two_passes_in_flight.jsonl 2> $O/two_passes.err
timeout 300 python tools/bench_latency.py > $O/latency.jsonl 2> $O/latency.err
# small batches: three lanes per item (the parted verify) against one, each leg its own process
timeout 300 python tools/parted_ab.py --once --reps=100 > $O/parted_verify_ab.jsonl 2> $O/parted_ab.err
# the N > 1 flow on the one-GPU box, started as PLAIN python (bench.py launches itself under
# torch.distributed.run): weak loop + configs[2] as written (`strong`) + the collective's census in
# one line; two ranks, then eight ranks, all on device 0 over gloo (flow tests, not scaling points)
( timeout 600 python3 bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --force-device 0 --sustain 1 ) > $O/bench_selflaunch_two_ranks_one_gpu.jsonl 2> $O/bench_selflaunch2.err
( timeout 900 python3 bench.py --gpus 8 --steps 6 --warmup 2 --dist-backend gloo --force-device 0 --batch 131072 --sustain 1 ) > $O/bench_selflaunch_eight_ranks_one_gpu.jsonl 2> $O/bench_selflaunch8.err
( timeout 300 python bench.py --rccl-selftest --steps 5 --warmup 2 --no-cpu --no-configs --no-live-counters --sustain 0 ) > $O/rccl_selftest.jsonl 2> $O/rccl_selftest.err
if [ "$MODE" != "quick" ]; then
  timeout 300 python tools/gpu_probe.py > $O/valu_probe.log 2>&1
  timeout 200 tools/microbench/_build/valu_patterns > $O/valu_patterns.log 2>&1
  timeout 200 tools/microbench/_build/u29_probe > $O/u29_probe.log 2>&1
  timeout 600 python tools/bench_configs.py 2>/dev/null | grep '"config"' > $O/configs.jsonl
  timeout 300 python tools/bench_host_path.py --reps 8 2>/dev/null | grep '"config"' > $O/host_path.jsonl
  timeout 120 node elliptic_amd/js/bench.js 2>/dev/null | grep '^{' > $O/js_bench.jsonl
  timeout 120 node tools/bench_js_single_call.js 2>/dev/null | grep '^{' > $O/js_single_call.jsonl
  timeout 300 python tools/bench_custom.py 18 > $O/custom_curve_bench.jsonl 2>/dev/null
  timeout 300 python tests/soak.py --seconds 40 > $O/soak.log 2>&1
  # round 6: the one-item-per-row gate, the wide layer's chains, where a lone call spends its time,
  # and the one-item-per-row window switched on and off (each leg its own process)
  timeout 300 python tools/microbench/row_items.py $O/row_items_gate.jsonl > $O/row_items_gate.log 2>&1
  timeout 300 python tools/microbench/wide_field.py $O/wide_field.jsonl > $O/wide_field.log 2>&1
  timeout 300 python tools/single_call_breakdown.py > $O/single_call_breakdown.jsonl 2> $O/single_call_breakdown.err
  export ELLGPU_LATENCY_SIZES="256,512,640,768,1024,1366,2048,3072,4096,4608,6144,8192,16384"
  timeout 300 python tools/bench_latency.py > $O/latency_rows_on.jsonl 2> $O/latency_rows.err
  ELLGPU_ROW_GRID=0 timeout 300 python tools/bench_latency.py > $O/latency_rows_off.jsonl 2>> $O/latency_rows.err
  unset ELLGPU_LATENCY_SIZES
  # the seeded differential fuzz of the public API (unpatched against patched reference) on the device,
  # and a longer soak against the C oracle
  for seed in r6-gpu-a r6-gpu-b r6-gpu-c; do timeout 600 node tools/fuzz_patched_vs_plain.js 600 $seed 2>&1 | tail -c 1200; echo; done > $O/fuzz_gpu.log
  timeout 400 python tests/soak.py --seconds 150 > $O/soak_long.log 2>&1
  # everything reachable from a protocol call's objects, changed one property at a time after first use
  # (four families side by side) and the kernel times of mid-size calls
  for f in short:secp256k1 short:p256 edwards:ed25519 mont:curve25519; do
    ( WALK_LIMIT_MS=20000 timeout 1500 node tools/probe_mutation_walk.js $f > $O/walk_${f#*:}.log 2>&1 ) &
  done
  wait
  cat $O/walk_secp256k1.log $O/walk_p256.log $O/walk_ed25519.log $O/walk_curve25519.log > $O/mutation_walk.log
  timeout 300 python tools/mid_batch_breakdown.py > $O/mid_batch_breakdown.jsonl 2>/dev/null
  python - > $O/latency_rows_ab.txt <<'PY'
import json, os
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "refresh")
def load(f):
    d = {}
    for l in open(os.path.join(O, f)):
        o = json.loads(l); d[(o["op"], o["n"])] = o["median_us"]
    return d
a, b = load("latency_rows_on.jsonl"), load("latency_rows_off.jsonl")
print("# host-buffer call latency, median of 30, us: default thresholds (one item per row for 641 .. 4 608 items) against ELLGPU_ROW_GRID=0")
for k in sorted(a, key=lambda k: (k[0], k[1])):
    print("%-14s %6d   rows on %8.1f   rows off %8.1f" % (k[0], k[1], a[k], b.get(k, 0)))
PY
fi
ls -la $O
