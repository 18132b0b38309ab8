O=gpurun_out/r03o; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tools/strong_proxy.py > $O/strong_proxy.jsonl 2> $O/strong_proxy.err; cut -c1-230 $O/strong_proxy.jsonl
