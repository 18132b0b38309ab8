#!/usr/bin/env python3
"""Does a second pass in flight hide the latency-bound front of a small pass?  One GPU's share of
2^20 over eight (131 072 verifies) spends 0.22 of its 1.30 ms in ecdsa_prep_table, a chain the
issue-bound ladder cannot hide inside ONE pass.  Here alternate passes go to two streams of ONE
context -- it gives each stream a scratch arena of its own (HipBackend::use_stream_dev) -- so that
pass i + 1's prep / tables run beside pass i's ladder.
    python tools/two_stream_probe.py
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import elliptic_amd


def main():
    ca = elliptic_amd.Context(0)
    cb = ca
    n0 = 1 << 20
    h, r, s, pub, expect = bench.cached_signatures(ca, n0, "ellgpu-bench-v1:3:rank0")
    dev = torch.device("cuda", 0)
    dh, dr, dsg, dq = (torch.from_numpy(x).to(dev) for x in (h, r, s, pub))
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for n in (1 << 20, 262144, 131072, 65536, 32768):
        oka = torch.zeros(n, dtype=torch.uint8, device=dev)
        okb = torch.zeros(n, dtype=torch.uint8, device=dev)
        args = (dh[:n], dr[:n], dsg[:n], dq[:n])
        reps = 60
        def one_stream():
            with torch.cuda.stream(sa):
                for _ in range(reps):
                    ca.ecdsa_verify_dev("secp256k1", *args, oka)
        def two_streams():
            for i in range(reps):
                if i % 2 == 0:
                    with torch.cuda.stream(sa):
                        ca.ecdsa_verify_dev("secp256k1", *args, oka)
                else:
                    with torch.cuda.stream(sb):
                        cb.ecdsa_verify_dev("secp256k1", *args, okb)
        out = {"n": n}
        for name, fn in (("one_stream", one_stream), ("two_streams", two_streams), ("one_stream_again", one_stream)):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            out[name + "_ms_per_pass"] = round((time.perf_counter() - t0) / reps * 1e3, 4)
        out["mask_ok"] = bool(np.array_equal(oka.cpu().numpy(), expect[:n]) and np.array_equal(okb.cpu().numpy(), expect[:n]))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
