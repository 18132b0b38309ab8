mkdir -p gpurun_out/c16
for ws in "3 2" "3 1" "2 2" "2 3" "2 4" "4 1" "1 4" "1 6" "3 2"; do set -- $ws
  echo "== waves $1 step $2" >> gpurun_out/c16/host.log
  ELLGPU_PIPE_WAVES=$1 ELLGPU_PIPE_STEP=$2 timeout 300 python tools/bench_host_path.py --reps 8 2>/dev/null | grep '"config"' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['config'][:40], round(d['items_per_s_best']/1e6,1), round(d['items_per_s_mean']/1e6,1))" >> gpurun_out/c16/host.log
done
cat gpurun_out/c16/host.log
