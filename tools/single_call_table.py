#!/usr/bin/env python3
"""Markdown rows "call | reference | patched | ratio" from a run of tools/bench_js_single_call.js
(profiles/rNN_js_single_call.jsonl): the table of INTEGRATION.md section 2.

    python tools/single_call_table.py profiles/r06_js_single_call.jsonl"""
import json
import sys


def main():
    ref, pat, order = {}, {}, []
    for line in open(sys.argv[1]):
        line = line.strip()
        if not line.startswith("{"):
            continue
        o = json.loads(line)
        (ref if o["library"].startswith("reference") else pat)[o["op"]] = o
        if o["op"] not in order:
            order.append(o["op"])
    print("| call | reference | patched | |")
    print("|---|---|---|---|")
    for op in order:
        p = pat.get(op)
        r = ref.get(op)
        if p is None:
            continue
        if r is None:
            extra = " (%s us per verification)" % p["per_verify_us"] if "per_verify_us" in p else ""
            print("| `%s` | | %s | %s |" % (op, ("%.0f" % p["median_us"]), extra.strip()))
        else:
            print("| `%s` | %.0f | **%.0f** | x%.1f |" % (op, r["median_us"], p["median_us"], r["median_us"] / p["median_us"]))


if __name__ == "__main__":
    main()
