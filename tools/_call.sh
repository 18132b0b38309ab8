O=gpurun_out/r03j; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $O/pytest_gpu.log 2>&1; tail -16 $O/pytest_gpu.log
python - <<'PY' > gpurun_out/r03j/comb_build_time.txt 2>&1
import time, numpy as np, elliptic_amd
ctx = elliptic_amd.Context(0)
for curve in ("secp256k1", "p256", "p192", "p384"):
    B = elliptic_amd.FIELD_BYTES[curve]
    k = np.ones((64, B), np.uint8)
    t0 = time.perf_counter(); ctx.mul_fixed(curve, k); t1 = time.perf_counter(); ctx.mul_fixed(curve, k); t2 = time.perf_counter()
    print(curve, "first mul_fixed (builds the comb): %.3f s, second: %.4f s" % (t1 - t0, t2 - t1))
PY
cat gpurun_out/r03j/comb_build_time.txt
