import os, sys, time, json
sys.path.insert(0, os.getcwd())
import numpy as np
import bench, elliptic_amd
ctx0 = elliptic_amd.Context(0)
h, r, s, pub, ok = bench.make_signatures(ctx0, 8192, "sweep")
res = []
for grid in ("0", str(1 << 30)):
    os.environ["ELLGPU_COOP_GRID"] = grid
    c = elliptic_amd.Context(0)
    for n in (512, 1024, 1365, 1536, 1792, 2048, 2560, 3072, 4096, 6144, 8192):
        for _ in range(3): c.ecdsa_verify("secp256k1", h[:n], r[:n], s[:n], pub[:n])
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); v = c.ecdsa_verify("secp256k1", h[:n], r[:n], s[:n], pub[:n]); ts.append(time.perf_counter() - t0)
        assert np.array_equal(v, ok[:n])
        ts.sort(); res.append({"coop": grid != "0", "n": n, "median_us": round(ts[len(ts) // 2] * 1e6, 1)})
        print(json.dumps(res[-1]), flush=True)
        kk = h[:n]
        for _ in range(2): c.mul_var("secp256k1", kk, pub[:n])
        ts = []
        for _ in range(9):
            t0 = time.perf_counter(); c.mul_var("secp256k1", kk, pub[:n]); ts.append(time.perf_counter() - t0)
        ts.sort(); print(json.dumps({"coop": grid != "0", "n": n, "mul_var_median_us": round(ts[len(ts) // 2] * 1e6, 1)}), flush=True)
    c.close()
