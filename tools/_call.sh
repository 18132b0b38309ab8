mkdir -p gpurun_out/c14
for l in ab_libs/a_new.so ab_libs/b_fixed4.so ab_libs/a_new.so ab_libs/b_fixed4.so; do
  ELLGPU_LIB=$l timeout 900 python tools/bench_configs.py --reps 5 --curves secp256k1,p192,p224,p256 2>/dev/null | grep '"config"' >> gpurun_out/c14/configs_$(basename $l).jsonl
done
python - <<'PY'
import json,glob
rows={}
for f in sorted(glob.glob('gpurun_out/c14/configs_*.jsonl')):
    for l in open(f):
        d=json.loads(l); rows.setdefault(d['config'],{}).setdefault(f.split('configs_')[1][:7],[]).append(d)
for k,v in rows.items():
    if 'fixed' in k or 'sign' in k:
        print(k[:64].ljust(64), {lib:[(round(d['items_per_s']/1e6,1), round(d['kernels_ms'].get('mul_fixed', d['kernels_ms'].get('sign_mul',0)),4)) for d in ds] for lib,ds in v.items()})
PY
