#!/bin/bash
# Runs on the GPU box (via gpurun): every measurement DESIGN.md section 6 cites, written under
# gpurun_out/refresh/.  `python tools/refresh_profiles.py` then copies the summaries into
# profiles/ with the round prefix.  PMC counters are collected in their own passes.
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/refresh
ONLY="$1"                      # optional: "pmc_fw" re-runs only the fetch/write passes
if [ "$ONLY" = "pmc_fw" ]; then
  mkdir -p $O
  timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fwa -o fwa -- python bench.py --steps 2 --warmup 1 > $O/pmc_fwa.log 2>&1
  timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_fwb -o fwb -- python bench.py --steps 2 --warmup 1 > $O/pmc_fwb.log 2>&1
  for t in fwa fwb; do
    db=$(find $O/prof_$t -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocprof_summary.py "$db" > $O/rocprof_$t.txt 2>&1
  done
  cat $O/rocprof_fwa.txt $O/rocprof_fwb.txt > $O/rocprof_fw.txt
  rm -rf $O/prof_fwa $O/prof_fwb
  ls -la $O
  exit 0
fi
rm -rf $O && mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -3 > $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench.json.log 2> $O/bench.err
timeout 300 python tools/gpu_probe.py > $O/valu_probe.log 2>&1
timeout 200 tools/microbench/_build/valu_patterns > $O/valu_patterns.log 2>&1
timeout 600 python tools/bench_configs.py 2>/dev/null | grep '"config"' > $O/configs.jsonl
timeout 300 python tools/bench_host_path.py --reps 8 2>/dev/null | grep '"config"' > $O/host_path.jsonl
timeout 300 python tools/bench_latency.py 2>/dev/null | grep '^{' > $O/latency.jsonl
timeout 120 node elliptic_amd/js/bench.js 2>/dev/null | grep '^{' > $O/js_bench.jsonl
timeout 120 node elliptic_amd/js/selftest.js > $O/js_selftest.log 2>&1
timeout 150 rocprofv3 --kernel-trace --stats -d $O/prof_stats -o stats -- python bench.py > $O/bench_under_rocprof.log 2>&1
# FETCH_SIZE and WRITE_SIZE do not fit one pass ("exceeds the capabilities of the hardware")
timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/prof_fwa -o fwa -- python bench.py --steps 2 --warmup 1 > $O/pmc_fwa.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/prof_fwb -o fwb -- python bench.py --steps 2 --warmup 1 > $O/pmc_fwb.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_ANY -d $O/prof_sqa -o sqa -- python bench.py --steps 2 --warmup 1 > $O/pmc_sqa.log 2>&1
timeout 150 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_IFETCH -d $O/prof_sqb -o sqb -- python bench.py --steps 2 --warmup 1 > $O/pmc_sqb.log 2>&1
for t in stats fwa fwb sqa sqb; do
  db=$(find $O/prof_$t -name "*_results.db" | head -1)
  [ -n "$db" ] && python tools/rocprof_summary.py "$db" > $O/rocprof_$t.txt 2>&1
done
cat $O/rocprof_fwa.txt $O/rocprof_fwb.txt > $O/rocprof_fw.txt
rm -rf $O/prof_stats $O/prof_fwa $O/prof_fwb $O/prof_sqa $O/prof_sqb
ls -la $O
