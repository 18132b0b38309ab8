#!/usr/bin/env python3
"""Copy the summaries tools/refresh_profiles.sh produced on the GPU box (gpurun_out/refresh/)
into profiles/ under the round prefix, and distil profiles/<round>_kernel_counters.json -- the
per-kernel instruction counts, VALU-busy and HBM traffic bench.py's `roofline` block reads.

    gpurun -- 'bash tools/refresh_profiles.sh' && python tools/refresh_profiles.py --round r02
"""
import argparse
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAP = {
    "pytest_gpu.log": "pytest_gpu.log",
    "bench.json.log": "bench.json.log",
    "valu_probe.log": "valu_probe.log",
    "valu_patterns.log": "valu_patterns.log",
    "u29_probe.log": "u29_probe.log",
    "configs.jsonl": "configs.jsonl",
    "host_path.jsonl": "host_path.jsonl",
    "js_bench.jsonl": "js_bench.jsonl",
    "js_single_call.jsonl": "js_single_call.jsonl",
    "soak.log": "soak.log",
    "custom_curve_bench.jsonl": "custom_curve_bench.jsonl",
    "bench_under_rocprof.log": "bench_under_rocprof.log",
    "rocprof_stats.txt": "rocprof_kernel_stats.txt",
    "rocprof_stats2.txt": "rocprof_kernel_stats_two_in_flight.txt",
    "bench_under_rocprof_two_in_flight.log": "bench_under_rocprof_two_in_flight.log",
    "two_passes_in_flight.jsonl": "two_passes_in_flight.jsonl",
    "rocprof_fw.txt": "rocprof_pmc_fetch_write.txt",
    "rocprof_sqa.txt": "rocprof_pmc_sq_a.txt",
    "rocprof_sqb.txt": "rocprof_pmc_sq_b.txt",
    "rocprof_gc.txt": "rocprof_pmc_gather_calib.txt",
    "gather_calib.log": "gather_calib.log",
    "smoke.log": "smoke.log",
    "strong_proxy.jsonl": "strong_proxy.jsonl",
    "latency.jsonl": "latency.jsonl",
    "parted_verify_ab.jsonl": "parted_verify_ab.jsonl",
    "bench_selflaunch_two_ranks_one_gpu.jsonl": "bench_selflaunch_two_ranks_one_gpu.jsonl",
    "bench_selflaunch_eight_ranks_one_gpu.jsonl": "bench_selflaunch_eight_ranks_one_gpu.jsonl",
    "rccl_selftest.jsonl": "rccl_selftest.jsonl",
    "row_items_gate.jsonl": "row_items_gate.jsonl",
    "wide_field.jsonl": "wide_field.jsonl",
    "single_call_breakdown.jsonl": "single_call_breakdown.jsonl",
    "latency_rows_ab.txt": "latency_rows_ab.txt",
    "fuzz_gpu.log": "fuzz_gpu.log",
    "soak_long.log": "soak_long.log",
    "mutation_walk.log": "mutation_walk.log",
    "mid_batch_breakdown.jsonl": "mid_batch_breakdown.jsonl",
}

# kernel name in the summaries -> (key bench.py uses, units per dispatch in the profiled command)
N20, N18 = 1 << 20, 1 << 18
KERNELS = {
    "k_run<FnEcdsaMain<CvSecp256k1>>": ("ecdsa_main<secp256k1>", N20),
    "k_run<FnEcdsaMain<CvSecp256k1>,wide>": ("ecdsa_main_small_grid<secp256k1>", None),   # host-buffer leg's first chunk
    "k_run<FnMulVar<CvSecp256k1>>": ("mul_var<secp256k1>", N20),
    "k_run<FnMulVar<CvSecp256k1>,wide>": ("mul_var_small_grid<secp256k1>", None),
    "k_run<FnMulFixed<CvSecp256k1>>": ("mul_fixed<secp256k1>", None),      # several grid sizes: per-lane figures only
    "k_run<FnMulVar<CvNist>>": ("mul_var<p384>", N18),        # the only NIST curve bench.py runs (names collapse to CvNist)
    "k_run<FnEdMulVar>": ("ed_mul_var", N20),
}


def counters(txt):
    """{kernel: {counter: (sum, dispatches)}} from a rocprof_summary.py text"""
    out = {}
    for m in re.finditer(r"^(\S.*?)\s+([A-Z][A-Z0-9_]+)\s+(\d+)\s+n=(\d+)\s+per_dispatch=(\d+)\s*$", txt, re.M):
        out.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), int(m.group(4)))
    return out


def calls(txt):
    """{kernel: dispatches} from the kernel-stats table of a summary"""
    out = {}
    sec = txt.split("## durations per dispatch")[0]
    for m in re.finditer(r"^(\S.*?)\s{2,}(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", sec, re.M):
        out[m.group(1).strip()] = int(m.group(2))
    return out


def totals(txt):
    """{kernel: total microseconds over all its dispatches} from the kernel-stats table of a summary"""
    out = {}
    sec = txt.split("## durations per dispatch")[0]
    for m in re.finditer(r"^(\S.*?)\s{2,}(\d+)\s+(\d+)\s+(\d+)\s+([\d.]+)\s*$", sec, re.M):
        out[m.group(1).strip()] = int(m.group(3))
    return out


def grids(txt):
    """{kernel: total work-items over all dispatches} from the dispatch table (first grid only per kernel)"""
    out = {}
    for m in re.finditer(r"^(\S.*?)\s+grid=(\d+)\s+wg=", txt, re.M):
        out.setdefault(m.group(1).strip(), []).append(int(m.group(2)))
    return out


def distil(src, digest):
    def read(name):
        p = os.path.join(src, name)
        return open(p).read() if os.path.exists(p) else ""
    sqa, sqb, fw, gc = read("rocprof_sqa.txt"), read("rocprof_sqb.txt"), read("rocprof_fw.txt"), read("rocprof_gc.txt")
    ca, cb, cf = counters(sqa), counters(sqb), counters(fw)
    na, nb = calls(sqa), calls(sqb)
    ta = totals(sqa)
    out = {"source_digest": digest, "how": "rocprofv3 --kernel-trace --pmc (separate passes) over "
           "`python bench.py --steps 2 --warmup 1 --no-cpu`; tools/refresh_profiles.sh",
           "kernels": {}}
    # FETCH_SIZE / WRITE_SIZE calibration on a known-bytes gather / scatter in this engine's pattern
    cal = None
    m = re.search(r"\{.*gather_bytes_per_dispatch.*\}", read("gather_calib.log"))
    if m and gc:
        known = json.loads(m.group(0))
        cg = counters(gc)
        for kname, c in cg.items():
            if "k_gather64" in kname and "FETCH_SIZE" in c:
                per = c["FETCH_SIZE"][0] / c["FETCH_SIZE"][1] * 1024.0     # KiB -> bytes per dispatch
                cal = {"known_gather_bytes": known["gather_bytes_per_dispatch"], "fetch_size_bytes": per,
                       "true_bytes_per_counted_byte": known["gather_bytes_per_dispatch"] / per if per else None,
                       "pattern": "64 B per lane (4 x dwordx4) at random 64-B-aligned offsets in a 2 GiB table"}
    out["fetch_calibration"] = cal
    for kname, (key, units) in KERNELS.items():
        a, b, f = ca.get(kname, {}), cb.get(kname, {}), cf.get(kname, {})
        if "SQ_INSTS_VALU" not in a or "SQ_WAVES" not in b:
            continue
        # Every pass runs the same command, so a kernel's dispatches (of whatever grid sizes: the
        # host-buffer leg cuts its batch into chunks) add up to the same number of wavefronts in
        # every pass.  SQ_INSTS_* count wavefront instructions: total / total waves = what ONE
        # lane (one verify / one scalar multiplication) executes.
        waves = float(b["SQ_WAVES"][0])
        ent = {"waves_in_pass": waves, "dispatches_in_pass": nb.get(kname)}
        ent["valu_per_unit"] = a["SQ_INSTS_VALU"][0] / waves
        ent["salu_per_unit"] = a.get("SQ_INSTS_SALU", (0, 1))[0] / waves
        if "SQ_INSTS_VALU_INT64" in b:
            ent["mad_u64_per_unit"] = b["SQ_INSTS_VALU_INT64"][0] / waves
            ent["int32_per_unit"] = b["SQ_INSTS_VALU_INT32"][0] / waves
        # VALU-busy: the gfx94x formulas rocprofv3 falls back to on gfx950
        if "SQ_ACTIVE_INST_VALU" in a and "SQ_BUSY_CYCLES" in a:
            act, busy = a["SQ_ACTIVE_INST_VALU"][0], a["SQ_BUSY_CYCLES"][0]
            gui = a.get("GRBM_GUI_ACTIVE", (0, 1))[0]
            ent["valu_busy"] = {
                "SQ_ACTIVE_INST_VALU": act, "SQ_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
                "SQ_THREAD_CYCLES_VALU": a.get("SQ_THREAD_CYCLES_VALU", (None,))[0],
                "SQ_WAVE_CYCLES": a.get("SQ_WAVE_CYCLES", (None,))[0],
                # VALUBusy (gfx94x derived metric) = 100 * SQ_ACTIVE_INST_VALU * 4 / SIMD_NUM / GRBM_GUI_ACTIVE
                # with the counters summed over the 32 SEs' SQs and GRBM over 8 XCDs as this summary has them:
                "valu_busy_pct_gfx94x_formula": (100.0 * act * 4 / 1024 / (gui / 8.0)) if gui else None,
                "active_inst_valu_over_busy_cycles": act / busy if busy else None,
                # effective clock of the kernel IN THE PROFILED PASS (MI355X_MICROARCH.md, DVFS note):
                # GRBM_GUI_ACTIVE (per XCD: the sum over the 8 XCDs / 8) / the kernel's wall time in the
                # same pass (the kernel trace of the pass that collected the counter)
                "kernel_total_us_in_pass": ta.get(kname),
                "clock_ghz_effective": (gui / 8.0 / ta[kname] / 1e3) if gui and ta.get(kname) else None,
            }
        if "FETCH_SIZE" in f and "WRITE_SIZE" in f:
            lanes = waves * 64.0
            fb = f["FETCH_SIZE"][0] * 1024.0                       # KiB, summed over the pass
            wb = f["WRITE_SIZE"][0] * 1024.0
            k = cal["true_bytes_per_counted_byte"] if cal and cal.get("true_bytes_per_counted_byte") else 1.0
            ent["fetch_bytes_per_unit_raw"] = fb / lanes
            ent["fetch_bytes_per_unit"] = fb * k / lanes
            ent["write_bytes_per_unit"] = wb / lanes
        out["kernels"][key] = ent
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r03")
    ap.add_argument("--src", default=os.path.join(ROOT, "gpurun_out", "refresh"))
    a = ap.parse_args()
    dst = os.path.join(ROOT, "profiles")
    for s, d in MAP.items():
        p = os.path.join(a.src, s)
        if not os.path.exists(p) or os.path.getsize(p) == 0:
            print("missing or empty:", s)
            continue
        shutil.copyfile(p, os.path.join(dst, "%s_%s" % (a.round, d)))
        print("profiles/%s_%s" % (a.round, d))
    sys.path.insert(0, ROOT)
    from elliptic_amd import build as _b
    digest = _b.library_digest()          # read from the built binary itself
    kc = distil(a.src, digest)
    if kc["kernels"]:
        with open(os.path.join(dst, "%s_kernel_counters.json" % a.round), "w") as f:
            json.dump(kc, f, indent=1)
        print("profiles/%s_kernel_counters.json: %s" % (a.round, ", ".join(kc["kernels"])))
    else:
        print("no PMC counters found: kernel_counters.json not written")


if __name__ == "__main__":
    sys.exit(main())
