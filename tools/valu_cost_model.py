#!/usr/bin/env python3
"""How busy are the vector ALUs?  One number instead of a [all-2-cycle, all-4-cycle] range.

No counter of gfx950 separates the VALU instructions by what they cost their SIMD (SQ_ACTIVE_INST_VALU counts in quad-cycles,
one per instruction: it cannot tell a 2-cycle v_fma_f32 from a 4-cycle v_pk_fma_f32).  So the cost is priced from the code:

  1. the ISA of the production kernel (render_wavefront_kernel<false, true, false, true, false, 256>, the arithmetic given) is split into
     its three kinds of wave-level work - NODE TRIP, LEAF PASS (both in the depth-2 traversal loop) and PATH-LOGIC PASS (the
     depth-1 remainder of the main loop) - and every vector instruction of each is put in a cost class with the cycle costs
     tools/microbench/pk_rate.hip measured on the MI355X: 8 for transcendentals (v_rcp / v_rsq / v_sqrt / v_sin / v_cos / v_exp /
     v_log), 4 for packed fp32 (v_pk_*), 64-bit forms, three-operand min / max / med, compares (they write an SGPR pair) and
     v_mbcnt / v_readlane / v_readfirstlane / v_writelane, 2 for every other 32-bit operation;
  2. each kind is weighted with how often a wave executes it: the trip counts of `bench.py --scheduler-stats`
     (trips_per_launch) - a wave runs the code of a trip for all its lanes, whatever the exec mask;
  3. the predicted instruction count (sum of trips x static count) is checked against the measured SQ_INSTS_VALU of the PMC
     passes, and the measured count is then priced at the model's mean cost per instruction:
         valu_busy = SQ_INSTS_VALU x mean_cycles_per_instruction / 1024 SIMDs / launch cycles.

Blocks that only run for rare rays (the literal 13-comparison box test of a wave holding a ray with a zero direction
component) are left out of a trip's static count; everything else counts once per trip - an over-estimate for the blocks a
trip skips when no lane needs them, which the check in 3. bounds.

usage: tools/valu_cost_model.py BENCH_JSON_WITH_SCHEDULER_STATS PMC_JSON [--arithmetic default|strict] [--general] > profiles/r04_valu_cost_model_<scene>.json
(--general: the general shading instantiation - textures, all material and light types - instead of the plain-scene one;
 --narrow: the instantiation with workgroups of 64 lanes, which deep trees run)
"""
import json
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD = 1024


def cost_class(op):
    if re.match(r"v_(rcp|rsq|sqrt|sin|cos|exp|log)_", op):
        return 8
    if op.startswith("v_pk_") or re.search(r"_(f64|i64|u64|b64)(_|$)", op) or re.match(r"v_(max3|min3|med3)_", op):
        return 4
    if op.startswith("v_cmp") or op.startswith("v_mbcnt") or op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return 4
    if op.startswith("v_mad_u64") or op.startswith("v_mad_i64"):
        return 4
    return 2


def kernel_blocks(arith, general=False, block=256):
    ns = "11ptmi_dev_da" if arith == "default" else "8ptmi_dev"
    asm = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
                          "-Iinclude", "-Iopencl_pathtracer_amd/csrc", f"-DPTMI_DEFAULT_ARITHMETIC={1 if arith == 'default' else 0}",
                          "--cuda-device-only", "-S", "opencl_pathtracer_amd/csrc/kernel_wavefront.hip", "-o", "-"],
                         cwd=ROOT, check=True, capture_output=True, text=True).stdout.split("\n")
    # general / plain-scene shading (what the workload's launches pick); <STATS, PRE, SS, PLAIN, NANSAFE>
    tag = f"ILb0ELb1ELb0ELb{0 if general else 1}ELb0ELi{block}EE"
    start = next(i for i, l in enumerate(asm) if l.startswith(f"_ZN{ns}23render_wavefront_kernel{tag}"))
    end = next(i for i in range(start, len(asm)) if asm[i].startswith(".Lfunc_end"))
    blocks, cur = [], None
    for l in asm[start:end]:
        if re.match(r"(\.LBB\d+_\d+:|; %bb\.\d+)", l):
            m = re.search(r"Depth=(\d)", l)
            cur = {"depth": int(m.group(1)) if m else 0, "ops": []}
            blocks.append(cur)
        elif cur is not None and re.match(r"\s+[a-z]", l):
            cur["ops"].append(l.split()[0])
    return blocks


def region_stats(blocks):
    c = Counter()
    for b in blocks:
        for op in b["ops"]:
            if op.startswith("v_"):
                c[cost_class(op)] += 1
    n = sum(c.values())
    cyc = sum(k * v for k, v in c.items())
    return {"valu_instructions": n, "by_cost_class": {str(k): c[k] for k in sorted(c)}, "issue_cycles": cyc,
            "mean_cycles_per_instruction": cyc / n if n else 0.0,
            "salu_instructions": sum(1 for b in blocks for op in b["ops"] if op.startswith("s_") and not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")))}


def main():
    bench = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    pmc = json.load(open(sys.argv[2]))
    arith = sys.argv[sys.argv.index("--arithmetic") + 1] if "--arithmetic" in sys.argv else bench.get("arithmetic", "default")
    general = "--general" in sys.argv
    block = 64 if "--narrow" in sys.argv else 256  # (workgroups of 64 lanes: what trees of 23 levels and more run)
    blocks = kernel_blocks(arith, general, block)
    loop2 = [b for b in blocks if b["depth"] == 2]
    # the traversal loop holds the node trip first, then the leaf pass: the pass starts at the block that builds the item
    # numbering (the only v_mbcnt of the loop) - the block before it is its scalar "is a pass due" test
    first_pass = next(i for i, b in enumerate(loop2) if any(op.startswith("v_mbcnt") for op in b["ops"]))
    node, leaf = loop2[:first_pass - 1], loop2[first_pass - 1:]
    # the rare block of a node trip: the literal box test (two boxes x 13 comparisons, no v_max3) between the ordered one and the push / pop
    rare = [b for b in node if len(b["ops"]) > 60 and not any(op.startswith("v_max3") for op in b["ops"])]
    node_common = [b for b in node if b not in rare]
    path = [b for b in blocks if b["depth"] == 1]
    regions = {"node_trip": region_stats(node_common), "leaf_pass": region_stats(leaf), "path_logic_pass": region_stats(path),
               "node_trip_rare_exact_box_test (not counted)": region_stats(rare)}
    trips = bench["wave_scheduler"]["trips_per_launch"]
    w = {"node_trip": trips["node"], "leaf_pass": trips["triangle"], "path_logic_pass": trips["path"]}
    pred_n = sum(w[k] * regions[k]["valu_instructions"] for k in w)
    pred_cyc = sum(w[k] * regions[k]["issue_cycles"] for k in w)
    mean = pred_cyc / pred_n
    v = lambda k: pmc[k]["per_launch_mean"]
    cycles = v("GRBM_GUI_ACTIVE") / 8.0
    measured_n = v("SQ_INSTS_VALU")
    out = {"kernel": f"render_wavefront_kernel<false,true,false,{'false' if general else 'true'},false,{block}> ({arith} arithmetic)", "workload": bench["config"]["workload"],
           "cycle_costs": "tools/microbench/pk_rate.hip on MI355X: 2 plain 32-bit, 4 packed fp32 / 64-bit / min3-max3 / compares / lane ops, 8 transcendental",
           "static_per_trip": regions, "trips_per_launch": w,
           "predicted_valu_instructions_per_launch": pred_n, "measured_SQ_INSTS_VALU_per_launch": measured_n,
           "predicted_over_measured": pred_n / measured_n,
           "share_of_valu_issue_cycles": {k: w[k] * regions[k]["issue_cycles"] / pred_cyc for k in w},
           "mean_cycles_per_valu_instruction": mean,
           "launch_cycles (GRBM_GUI_ACTIVE / 8 XCDs)": cycles,
           "valu_busy": measured_n * mean / N_SIMD / cycles,
           "valu_busy_if_all_2_cycles": measured_n * 2.0 / N_SIMD / cycles, "valu_busy_if_all_4_cycles": measured_n * 4.0 / N_SIMD / cycles,
           "scalar_unit_busy": v("SQ_INSTS_SALU") / 256.0 / cycles,
           "sources": {"bench": os.path.basename(sys.argv[1]), "pmc": os.path.basename(sys.argv[2]), "kernel_source_digest": pmc.get("_kernel_source_digest")}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
