#!/usr/bin/env python3
"""H2D / D2H rates for pageable, pinned and freshly allocated host memory on the GPU box
(background for Engine::pipelined).  GPU box only.

    python tools/copy_probe.py
"""
import time, sys, os
import numpy as np, torch
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
n = 64 << 20
d = torch.randint(0, 255, (n,), dtype=torch.uint8, device="cuda")
h = torch.empty(n, dtype=torch.uint8)
hp = torch.empty(n, dtype=torch.uint8).pin_memory()
hn = torch.from_numpy(np.zeros(n, np.uint8))
print("D2H 64MB pageable torch ms", t(lambda: h.copy_(d)))
print("D2H 64MB pageable numpy ms", t(lambda: hn.copy_(d)))
print("D2H 64MB pinned ms", t(lambda: hp.copy_(d, non_blocking=True)))
print("H2D 64MB pageable ms", t(lambda: d.copy_(h)))
print("H2D 64MB pinned ms", t(lambda: d.copy_(hp, non_blocking=True)))
print("host memcpy 64MB ms", t(lambda: h.copy_(hp)))
def fresh():
    x = torch.from_numpy(np.zeros(n, np.uint8)); x.copy_(d)
print("D2H 64MB fresh np.zeros ms", t(fresh))
