O=gpurun_out/r03h; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
