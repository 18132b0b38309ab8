mkdir -p gpurun_out/c6
for l in ab_libs/a_base.so elliptic_amd/lib/libellgpu.so; do
  ELLGPU_LIB=$l timeout 900 python tools/bench_configs.py --reps 3 > gpurun_out/c6/configs_$(basename $l).jsonl 2> gpurun_out/c6/err_$(basename $l).log
done
python - <<'PY'
import json,glob
rows={}
for f in sorted(glob.glob('gpurun_out/c6/configs_*.jsonl')):
    for l in open(f):
        if l.startswith('{'):
            d=json.loads(l); rows.setdefault((d.get('curve'),d.get('op')),{})[f.split('configs_')[1]]=d
for k,v in rows.items():
    print(k, {lib:(round(d.get('items_per_s',0)/1e6,2), d.get('kernels_ms')) for lib,d in v.items()})
PY
