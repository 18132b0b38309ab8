#!/usr/bin/env python3
"""Strong-scaling proxy on ONE GPU: BASELINE.json configs[2] is "batch = 1M sharded 1 -> 8
MI355X", i.e. every GPU of an N-GPU run verifies 2^20 / N tuples.  This measures what one GPU
does with such a shard -- ellgpu_ecdsa_verify_dev at n = 2^20 / {1, 2, 4, 8} (and two smaller
sizes) -- per-kernel HIP-event times, wall time per pass, and the per-item efficiency against
the full batch.  predicted_speedup(N) = t(2^20) / t(2^20 / N): what N GPUs would give before
the gather.  GPU box only.

    python tools/strong_proxy.py [--reps 40] > gpurun_out/strong_proxy.jsonl
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import elliptic_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--total", type=int, default=1 << 20)
    a = ap.parse_args()
    ctx = elliptic_amd.Context(0)
    n0 = a.total
    h, r, s, pub, expect = bench.cached_signatures(ctx, n0, "ellgpu-bench-v1:3:rank0")
    dev = torch.device("cuda", 0)
    dh, dr, dsg, dq = (torch.from_numpy(x).to(dev) for x in (h, r, s, pub))
    ctx.reserve("secp256k1", n0)
    rows = []
    for div in (1, 2, 4, 8, 16, 64):
        n = n0 // div
        dok = torch.zeros(n, dtype=torch.uint8, device=dev)
        args = (dh[:n], dr[:n], dsg[:n], dq[:n], dok)
        for _ in range(5):
            ctx.ecdsa_verify_dev("secp256k1", *args)
        torch.cuda.synchronize()
        assert np.array_equal(dok.cpu().numpy(), expect[:n]), "parity at n=%d" % n
        # wall time per pass WITHOUT per-launch events (what a rank of bench.py sees) ...
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            ctx.ecdsa_verify_dev("secp256k1", *args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.reps
        # ... the same with TWO passes in flight (alternate passes on two streams, each with its own
        # result buffer and its own scratch arena of the context: bench.py's default) ...
        dok2 = torch.zeros(n, dtype=torch.uint8, device=dev)
        lanes = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
        outs = [dok, dok2]
        torch.cuda.synchronize()
        for warm in (True, False):
            t0 = time.perf_counter()
            for i in range(a.reps):
                with torch.cuda.stream(lanes[i & 1]):
                    ctx.ecdsa_verify_dev("secp256k1", *args[:4], outs[i & 1])
            torch.cuda.synchronize()
            dt2 = (time.perf_counter() - t0) / a.reps
        assert np.array_equal(dok2.cpu().numpy(), expect[:n]) and np.array_equal(dok.cpu().numpy(), expect[:n]), "parity, two in flight, n=%d" % n
        # ... then the per-kernel breakdown with HIP events around every launch
        ctx.set_timing(True)
        for _ in range(a.reps):
            ctx.ecdsa_verify_dev("secp256k1", *args)
        torch.cuda.synchronize()
        tm = ctx.get_timing()
        ctx.set_timing(False)
        rows.append({"n": n, "shards": div, "ms_per_pass": dt * 1e3, "ms_per_pass_two_in_flight": dt2 * 1e3,
                     "kernels_ms": {k: v[1] / max(v[0], 1) for k, v in tm.items()},
                     "ns_per_item": dt * 1e9 / n, "library_digest": bench.lib_digest()})
    base = rows[0]
    for row in rows:
        row["per_item_efficiency"] = base["ns_per_item"] / row["ns_per_item"]
        row["predicted_speedup_before_gather"] = base["ms_per_pass"] / row["ms_per_pass"]
        row["predicted_speedup_two_in_flight"] = base["ms_per_pass_two_in_flight"] / row["ms_per_pass_two_in_flight"]
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
