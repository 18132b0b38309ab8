#!/usr/bin/env python3
"""Copy the summaries tools/refresh_profiles.sh produced on the GPU box (gpurun_out/refresh/)
into profiles/ under the round prefix.

    gpurun -- 'bash tools/refresh_profiles.sh' && python tools/refresh_profiles.py [--round r01]
"""
import argparse
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAP = {
    "pytest_gpu.log": "pytest_gpu.log",
    "bench.json.log": "bench.json.log",
    "valu_probe.log": "valu_probe.log",
    "valu_patterns.log": "valu_patterns.log",
    "configs.jsonl": "configs.jsonl",
    "host_path.jsonl": "host_path.jsonl",
    "latency.jsonl": "latency.jsonl",
    "js_bench.jsonl": "js_bench.jsonl",
    "js_selftest.log": "js_selftest.log",
    "bench_under_rocprof.log": "bench_under_rocprof.log",
    "rocprof_stats.txt": "rocprof_kernel_stats.txt",
    "rocprof_fw.txt": "rocprof_pmc_fetch_write.txt",
    "rocprof_sqa.txt": "rocprof_pmc_sq_a.txt",
    "rocprof_sqb.txt": "rocprof_pmc_sq_b.txt",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    a = ap.parse_args()
    src = os.path.join(ROOT, "gpurun_out", "refresh")
    dst = os.path.join(ROOT, "profiles")
    for s, d in MAP.items():
        p = os.path.join(src, s)
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            print("missing or empty:", s)
            continue
        shutil.copyfile(p, os.path.join(dst, "%s_%s" % (a.round, d)))
        print("profiles/%s_%s" % (a.round, d))


if __name__ == "__main__":
    main()
