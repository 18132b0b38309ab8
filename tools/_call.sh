set -x
mkdir -p gpurun_out/c1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_baseline_sizes.py tests/test_js_install.py -m gpu -x -q -k "user_defined or custom" > gpurun_out/c1/pytest_custom.log 2>&1
tail -5 gpurun_out/c1/pytest_custom.log
for l in elliptic_amd/lib/libellgpu.so ab_libs/custom_w2.so ab_libs/custom_w4.so; do ELLGPU_LIB=$l timeout 300 python tools/bench_custom.py 18 >> gpurun_out/c1/bench_custom.jsonl 2>gpurun_out/c1/bench_custom.err; done
cat gpurun_out/c1/bench_custom.jsonl
