mkdir -p gpurun_out/c7
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/c7/pytest_gpu.log 2>&1
tail -4 gpurun_out/c7/pytest_gpu.log
