#!/bin/bash
# Round-3 evidence, run from the repo root ON THE GPU BOX (everything lands in gpurun_out/, the summaries are copied to profiles/):
#   1. rocprofv3 kernel trace + stats and one --pmc pass per counter group of the default bench command (tools/profile_round.sh)
#   2. bench.py --scheduler-stats (trip counts)  ->  tools/valu_cost_model.py  (one VALU-busy number)
#   3. the bench line itself (default arithmetic, all legs) and the strict-arithmetic line
# usage: tools/r03_evidence.sh [tag]
TAG=${1:-r03}
bash tools/profile_round.sh $TAG --no-reference-kernel > gpurun_out/profile_round_$TAG.log 2>&1
python bench.py --steps 3 --warmup 1 --scheduler-stats --no-cpu-baseline --no-boundary --no-reference-kernel > gpurun_out/${TAG}_bench_scheduler_stats.json 2> gpurun_out/${TAG}_bench_scheduler_stats.err
python tools/valu_cost_model.py gpurun_out/${TAG}_bench_scheduler_stats.json gpurun_out/pmc_$TAG.json > gpurun_out/${TAG}_valu_cost_model.json 2> gpurun_out/${TAG}_valu_cost_model.err
mkdir -p profiles
cp gpurun_out/pmc_$TAG.json profiles/r03_pmc_tris1m_default.json
cp gpurun_out/${TAG}_valu_cost_model.json profiles/r03_valu_cost_model_tris1m_default.json
python bench.py --steps 10 --warmup 2 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 10 --warmup 2 --arithmetic strict --no-cpu-baseline --no-boundary > gpurun_out/${TAG}_bench_strict.json 2> gpurun_out/${TAG}_bench_strict.err
# BASELINE configs[4]'s stand-in: PMC passes + scheduler statistics + the bench line
MM="--scene matmix --width 3840 --height 2160 --depth 16 --spp-per-step 25"   # (25: the most a 4K launch takes, so one step = one launch)
bash tools/profile_round.sh ${TAG}_matmix --no-reference-kernel $MM > gpurun_out/profile_round_${TAG}_matmix.log 2>&1
python bench.py --steps 3 --warmup 1 --scheduler-stats --no-cpu-baseline --no-boundary --no-reference-kernel $MM > gpurun_out/${TAG}_bench_scheduler_stats_matmix.json 2> gpurun_out/${TAG}_bench_scheduler_stats_matmix.err
python tools/valu_cost_model.py gpurun_out/${TAG}_bench_scheduler_stats_matmix.json gpurun_out/pmc_${TAG}_matmix.json --general > gpurun_out/${TAG}_valu_cost_model_matmix.json 2>> gpurun_out/${TAG}_valu_cost_model.err
cp gpurun_out/pmc_${TAG}_matmix.json profiles/r03_pmc_matmix_default.json
cp gpurun_out/${TAG}_valu_cost_model_matmix.json profiles/r03_valu_cost_model_matmix_default.json
python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-reference-kernel $MM > gpurun_out/${TAG}_bench_matmix.json 2> gpurun_out/${TAG}_bench_matmix.err
tail -c 400 gpurun_out/${TAG}_valu_cost_model.err
python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print(round(d['value'],1), d['vs_baseline'], r['bound'], r['frac'], {k:(round(v['frac'],3) if 'frac' in v else None) for k,v in r.get('binding',{}).items() if isinstance(v,dict)})
PY
