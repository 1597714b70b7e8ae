#!/usr/bin/env python3
"""Joins the timing lines of tools/microbench/l1_gather with the TCP counters of the same launches (tools/l1_ceiling.sh) into
the record bench.py's roofline reads: profiles/r04_l1_gather_microbench.json."""
import csv
import glob
import json
import os
import sys

d = sys.argv[1]
out = {"what": "lane accesses per second the L1s (TCP) of one MI355X serve for the integrator's load pattern - every lane reads one "
               "64-byte record of its own with dwordx4 loads - in three regimes (tools/microbench/l1_gather.hip, tools/l1_ceiling.sh); "
               "pmc_* = rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum of the same launches",
       "regimes": {}}
for path in sorted(glob.glob(os.path.join(d, "*.jsonl"))):
    name = os.path.basename(path)[:-6]
    runs = [json.loads(l) for l in open(path) if l.startswith("{")]
    # counters: one dispatch per in-flight count, in the order 1, 2, 4, three repetitions each (the bench keeps the best time)
    per_dispatch = {}
    for f in glob.glob(os.path.join(d, "pmc_" + name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gather" in r["Kernel_Name"]:
                per_dispatch.setdefault(int(r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
                per_dispatch[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    ids = sorted(per_dispatch)
    for k, run in enumerate(runs):
        mine = [per_dispatch[i] for i in ids[3 * k:3 * k + 3]]
        if mine and all("TCP_TOTAL_CACHE_ACCESSES_sum" in m for m in mine):
            acc = sum(m["TCP_TOTAL_CACHE_ACCESSES_sum"] for m in mine) / len(mine)
            fills = sum(m.get("TCP_TCC_READ_REQ_sum", 0.0) for m in mine) / len(mine)
            quads = run["G_lane_accesses_per_s"] * 1e9 * run["ms"] * 1e-3
            run["pmc_TCP_TOTAL_CACHE_ACCESSES_per_launch"] = acc
            run["pmc_TCP_TCC_READ_REQ_per_launch"] = fills
            run["pmc_accesses_per_quad"] = round(acc / quads, 4)      # 1.0: the counter counts one access per lane per dwordx4
            run["pmc_fills_per_access"] = round(fills / acc, 4)
            run["pmc_G_accesses_per_s"] = round(acc / (run["ms"] * 1e-3) / 1e9, 2)
            run["pmc_G_fills_per_s"] = round(fills / (run["ms"] * 1e-3) / 1e9, 2)
    out["regimes"][name] = {"runs": runs}


def best(name):
    return max((r["G_lane_accesses_per_s"] for r in out["regimes"].get(name, {}).get("runs", [])), default=None)


out["ceiling"] = {"pure_access_rate_G_per_s": best("hit"),
                  "kernel_mix_G_per_s": best("kernel_mix"),
                  "kernel_mix_cold_from_infinity_cache_G_per_s": best("kernel_mix_cold_from_infinity_cache"),
                  "every_record_a_new_line_from_l2_G_per_s": best("every_record_a_new_line_l2"),
                  "every_record_a_new_line_G_per_s": best("every_record_a_new_line"),
                  "note": "kernel_mix = the regime of render_wavefront_kernel on the 1M-triangle workload by its own PMC profile (600 of "
                          "1000 records cost a line fill, served by the L2s - L2 hit rate 99 % -, 3.14 accesses per record): what bench.py "
                          "divides by, at the best in-flight count; pmc_fills_per_access tells which of the two L1 rates a regime was "
                          "limited by (0: accesses; ~0.25-0.5 with a flat access rate: line fills)"}
print(json.dumps(out, indent=1))
