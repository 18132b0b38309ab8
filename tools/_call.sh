cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r02d; mkdir -p $O
timeout 300 python tools/ab_variants.py ab_libs/a_lean_w3.so ab_libs/d_lambda_lookup.so --reps=30 > $O/ab.jsonl 2> $O/ab.err
cat $O/ab.jsonl | cut -c1-330
for v in a_lean_w3 d_lambda_lookup; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ELLGPU_LIB=$PWD/ab_libs/$v.so timeout 200 rocprofv3 --kernel-trace --pmc $c -d $O/prof_${v}_$c -o p -- python tools/ab_variants.py --child ab_libs/$v.so 1048576 3 > $O/pmc_${v}_$c.log 2>&1
    db=$(find $O/prof_${v}_$c -name "*_results.db" | head -1)
    [ -n "$db" ] && python tools/rocprof_summary.py "$db" | grep -E "FnEcdsaMain.*(FETCH|WRITE)_SIZE|FnMulVar.*(FETCH|WRITE)_SIZE" | sed "s/^/$v /"
    rm -rf $O/prof_${v}_$c
  done
done | tee $O/lambda_traffic.txt
for v in p384_w2 p384_w3; do
  echo "== $v"; ELLGPU_LIB=$PWD/ab_libs/$v.so timeout 200 python tools/bench_configs.py --curves p384 --reps 5 2>/dev/null | grep -E "variable-base|fixed-base|ECDSA verify\"" | cut -c1-200
done | tee $O/p384_waves.txt
