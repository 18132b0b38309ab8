import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench, elliptic_amd
ctx = elliptic_amd.Context(0)
n = 1 << 20
hz, hr, hs, hq, want = bench.make_signatures(ctx, n, "host-path")
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
x = np.random.randint(0, 255, 168 * (1 << 20), dtype=np.uint8)
tx = torch.from_numpy(x)
d = torch.empty_like(tx, device="cuda")
print("pageable H2D 168MB ms", t(lambda: d.copy_(tx)))
px = tx.pin_memory()
print("pinned H2D 168MB ms", t(lambda: d.copy_(px, non_blocking=True)))
for m in (196608, 393216, 589824, 786432, 1 << 20):
    print(m, "host verify ms", t(lambda: ctx.ecdsa_verify("secp256k1", hz[:m], hr[:m], hs[:m], hq[:m])))
dz, dr, ds_, dq = (torch.from_numpy(a).cuda() for a in (hz, hr, hs, hq))
ok = torch.empty(n, dtype=torch.uint8, device="cuda")
for m in (196608, 393216, 589824, 786432, 1 << 20):
    print(m, "dev verify ms", t(lambda: ctx.ecdsa_verify_dev("secp256k1", dz[:m], dr[:m], ds_[:m], dq[:m], ok[:m])))
