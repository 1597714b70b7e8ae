#!/bin/bash
# The ceiling of bench.py's `roofline` (L1 lane accesses per second), measured in the kernel's own regime and checked with the
# L1's own counters.  Run from the repo root ON THE GPU BOX; writes gpurun_out/r04_l1_gather_microbench.json (copy it to
# profiles/: bench.py reads profiles/r04_l1_gather_microbench.json).
#   regimes: pure access rate (every record an L1 hit), the kernel's mix (600 of 1000 records cost a line fill, 3.14 accesses
#   per record), every record a new line (what round 3's 858 G/s constant was taken from)
R=$PWD
B=$R/tools/microbench/l1_gather
[ -x $B ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/l1_gather.hip -o $B || exit 1
OUT=$R/gpurun_out/l1_gather
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (cold records: `kernel_mix` draws them from a 2 MB set - every line fill an L2 hit, as in the kernel, whose L2 hit rate is 99 % -,
# `kernel_mix_cold_from_infinity_cache` from 107 MB, `every_record_a_new_line` from 107 MB: what round 3's constant was taken on)
for regime in "hit 0 1000 1665533" "hit_kernel_quads 0 570 1665533" "kernel_mix 600 570 32768" "kernel_mix_cold_from_infinity_cache 600 570 1665533" "every_record_a_new_line_l2 1000 1000 32768" "every_record_a_new_line 1000 1000 1665533"; do
  set -- $regime
  name=$1; miss=$2; late=$3; cold=$4
  $B --miss-permille $miss --late-permille $late --cold-records $cold > $OUT/$name.jsonl || exit 1
  # the same three launches under the L1's counters (one pass, counters only)
  rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/pmc_$name -o c -- $B --miss-permille $miss --late-permille $late --cold-records $cold > $OUT/pmc_$name.log 2>&1 || echo "pmc pass of $name failed"
  echo "$name done"
done
cd $R
python3 tools/l1_ceiling_summary.py $OUT > gpurun_out/r04_l1_gather_microbench.json && python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r04_l1_gather_microbench.json"))
for name, r in d["regimes"].items():
    for e in r["runs"]:
        print(name, "in flight", e["in_flight_per_lane"], "G accesses/s", e["G_lane_accesses_per_s"], "G records/s", e["G_records_per_s"],
              "| counters: accesses per quad read", e.get("pmc_accesses_per_quad"), "fills per access", e.get("pmc_fills_per_access"))
print("ceiling", d["ceiling"])
PY
