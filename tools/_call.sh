O=gpurun_out/r03l; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python tools/strong_proxy.py > $O/strong_proxy.jsonl 2> $O/strong_proxy.err; cut -c1-260 $O/strong_proxy.jsonl
timeout 600 python tools/bench_configs.py --reps 4 2>/dev/null | grep '"config"' > $O/configs.jsonl
python - <<'PY'
import json
for l in open('gpurun_out/r03l/configs.jsonl'):
    r=json.loads(l)
    print(r['config'], round(r['items_per_s']/1e6,2))
PY
