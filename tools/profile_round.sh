#!/bin/bash
# Collect the rocprofv3 evidence bench.py's roofline refers to (run from the repo root ON THE GPU BOX):
#   1. kernel trace + stats of the bench command              -> gpurun_out/prof_$TAG/wf_kernel_stats.csv
#   2. one --pmc pass per counter group (never mixed with other trace domains) -> gpurun_out/pmc_$TAG/*.csv
#   3. tools/summarize_pmc.py                                -> gpurun_out/pmc_$TAG.json (copy to profiles/r04_pmc_<scene>_<arithmetic>.json)
# usage: tools/profile_round.sh TAG [bench.py flags, e.g. --scene tris4m]
set -e
TAG=${1:?tag}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o wf -- python3 $R/bench.py --no-cpu-baseline --no-boundary "$@" > $R/gpurun_out/bench_prof_$TAG.log 2>&1
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-20)
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$TAG -o $n -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-boundary "$@" > $R/gpurun_out/pmc_${TAG}_$n.log 2>&1 || echo "pmc pass '$c' failed"
  echo "pass $n done"
done
cd $R
python3 tools/summarize_pmc.py gpurun_out/pmc_$TAG "$*" > gpurun_out/pmc_$TAG.json
