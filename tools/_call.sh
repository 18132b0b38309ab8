O=gpurun_out/r03m; mkdir -p $O; rm -f $O/p384_pf.jsonl
for lib in p384_w3 p384_w3pf p384_w2pf p384_w3; do
  ELLGPU_LIB=$PWD/ab_libs/$lib.so timeout 600 python tools/bench_configs.py --reps 4 --curves p384 2>/dev/null | grep '"config"' | sed "s/^/{\"lib\": \"$lib\", \"row\": /; s/$/}/" >> $O/p384_pf.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r03m/p384_pf.jsonl'):
    d=json.loads(l); r=d['row']
    if 'variable-base' in r['config'] or 'verify' in r['config'] or 'fixed' in r['config']:
        print(d['lib'], r.get('config'), round(r['items_per_s']/1e6,2), {k:round(v,3) for k,v in r['kernels_ms'].items()})
PY
