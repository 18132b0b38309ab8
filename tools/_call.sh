O=gpurun_out/r03d; mkdir -p $O
( time timeout 900 python bench.py --keep-counters $O/live_counters ) > $O/bench.json.log 2> $O/bench.err; tail -c 300 $O/bench.err
( time timeout 900 python bench.py --steps 20 --warmup 3 --no-cpu --no-configs ) > $O/bench2.json.log 2> $O/bench2.err; tail -c 300 $O/bench2.err
ls $O/live_counters
