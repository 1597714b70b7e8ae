#!/bin/bash
# One rocprofv3 --pmc pass per argument (a quoted, space-separated counter group) over a short bench run; prints
# per-launch means of render_wavefront_kernel.  Run from the repo root ON THE GPU BOX.
# usage: tools/pmc_pass.sh TAG "CTR_A CTR_B" "CTR_C" ...
TAG=${1:?tag}; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcx_$TAG -o g$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline ${BENCH_ARGS} > $R/gpurun_out/pmcx_${TAG}_g$i.log 2>&1 || echo "pass '$c' failed: $(tail -2 $R/gpurun_out/pmcx_${TAG}_g$i.log | cut -c1-300)"
done
cd $R
python3 tools/summarize_pmc.py gpurun_out/pmcx_$TAG > gpurun_out/pmcx_$TAG.json
python3 - <<PY
import json; d=json.load(open('gpurun_out/pmcx_$TAG.json'))
for k,v in d.items():
    if k!='_note': print(k, '%.4g'%v['per_launch_mean'])
PY
