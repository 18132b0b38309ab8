O=gpurun_out/r03k; mkdir -p $O
python tools/ab_variants.py ab_libs/base.so ab_libs/glvodd.so --reps=30 > $O/glvodd_ab.jsonl 2> $O/ab.err
python - <<'PY'
import json
for l in open('gpurun_out/r03k/glvodd_ab.jsonl'):
    d=json.loads(l)
    print({k:d[k] for k in d if k in ('lib','mask_ok','ecdsa_main_ms','mul_var_same','mul_fixed_same','mul_fixed_kernel_ms','mul_var_kernel_ms','error')})
PY
