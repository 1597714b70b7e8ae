#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the summaries the round's numbers come from
set -e
G=gpurun_out; P=profiles
cp $G/r02_final_bench.json $P/r02_wavefront_bench.json
cp $G/r02_final_bench_sched.json $P/r02_wavefront_bench_scheduler_stats.json
cp $G/r02_final_ref_tris1m.json $P/r02_reference_kernel_on_mi355x_tris1m.json
cp $G/r02_final_ref_cornell.json $P/r02_reference_kernel_on_mi355x_cornell.json
for f in $G/r02_bench_config*.json $G/r02_bench_tris4m*.json; do cp $f $P/$(basename $f); done
cp $G/pmc_r02_tris1m.json $P/r02_pmc_tris1m.json
cp $G/pmc_r02_tris4m.json $P/r02_pmc_tris4m.json
cp $G/prof_r02_tris1m/wf_kernel_stats.csv $P/r02_wavefront_kernel_stats_tris1m.csv
cp $G/prof_r02_tris4m/wf_kernel_stats.csv $P/r02_wavefront_kernel_stats_tris4m.csv
cp $G/r02_north_star_rms.json $P/r02_north_star_rms.json
ls -la $P | grep r02
# (TCP_* counters: tools/profile_round.sh collects them as one more group; merged by key into the same json)
