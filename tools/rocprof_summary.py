#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite) into a text summary: per-kernel
stats (calls / total / average / %) and, when present, PMC counter sums per
kernel.   python tools/rocprof_summary.py <results.db> [> profiles/xxx.txt]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"k_run<ell::(\w+)(?:<ell::(\w+)([^()]*?)>)?", name)
    if m:
        # the small-grid (WIDE) instantiation of a functor is a kernel of its own
        wide = ",wide" if (m.group(3) or "").replace(" ", "").endswith(",true") else ""
        return "k_run<%s%s%s>" % (m.group(1), ("<" + m.group(2) + ">") if m.group(2) else "", wide)
    return name[:90]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 summary of", path)
    print("## kernel stats (durations in microseconds as rocprofv3 top_kernels reports them)")
    print("%-60s %6s %14s %14s %7s" % ("kernel", "calls", "total_us", "avg_us", "%"))
    for name, calls, total, avg, pct in cur.execute(
            "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        print("%-60s %6d %14d %14.0f %7.2f" % (short(name), calls, total, avg, pct))
    # per-dispatch durations in launch order: the first launches of a kernel (cold code / clocks,
    # bench.py's untimed warm-up steps) are slower than the steady state its HIP events time
    print("\n## durations per dispatch, launch order (ms)")
    per = {}
    for name, dur in cur.execute("select name,duration from kernels order by start"):
        per.setdefault(short(name), []).append(dur / 1e6)
    for name, ds in per.items():
        if len(ds) > 1 and max(ds) > 0.05:
            tail = ds[len(ds) // 2:]
            print("%-60s %s   | mean of the last %d: %.3f" % (name, " ".join("%.3f" % d for d in ds[:12]),
                                                            len(tail), sum(tail) / len(tail)))
    print("\n## dispatches (grid, workgroup, VGPR/AGPR/SGPR, LDS, scratch)")
    seen = set()
    for row in cur.execute("select name,grid_x,workgroup_x,vgpr_count,accum_vgpr_count,sgpr_count,lds_size,scratch_size,duration from kernels order by start"):
        key = (row[0], row[1])
        if key in seen:
            continue
        seen.add(key)
        print("%-60s grid=%-9d wg=%-4d vgpr=%-4s agpr=%-4s sgpr=%-4s lds=%-6s scratch=%-5s dur_ns=%d" % ((short(row[0]),) + row[1:]))
    try:
        rows = list(cur.execute(
            "select k.name, p.name, sum(e.value), count(*) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id "
            "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join kernels k on k.dispatch_id = d.dispatch_id "
            "group by k.name, p.name order by k.name, p.name"))
    except sqlite3.Error as ex:
        rows = []
        print("\n(no PMC data: %s)" % ex)
    if rows:
        print("\n## PMC counters (sum over dispatches; n = dispatches)")
        for kname, pname, val, n in rows:
            print("%-60s %-24s %20.0f  n=%d  per_dispatch=%.0f" % (short(kname), pname, val, n, val / n))


if __name__ == "__main__":
    main(sys.argv[1])
