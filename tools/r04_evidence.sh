#!/bin/bash
# Round-4 evidence, run from the repo root ON THE GPU BOX (everything lands in gpurun_out/; the summaries are copied to profiles/
# afterwards by hand: r04_pmc_<scene>_default.json, r04_valu_cost_model_<scene>_default.json, r04_wavefront_kernel_stats_<scene>.csv,
# r04_bench_*.json).  The PMC passes carry the digest of the kernel sources they were taken on (bench.kernel_source_digest):
# THIS IS THE LAST ACT OF A ROUND - any later edit of the four kernel files drops the bench line to its loud fallback.
#   part tris1m   : kernel trace + stats, one --pmc pass per counter group, scheduler statistics -> VALU cost model, the bench
#                   line with all its legs, the strict-arithmetic line
#   part mayalike : the same for the configs[4] stand-in (3840x2160, depth 16, 25 iterations per launch)
#   part cornell  : the same for BASELINE configs[1] (Cornell box 1080p, depth 8)
#   part configs  : BASELINE's other configs (tools/bench_configs.sh), the north-star parity check at full sizes and sample counts
# usage: tools/r04_evidence.sh tris1m|mayalike|cornell|configs
part=${1:?part}
mkdir -p gpurun_out
one_scene() {  # tag, general flag for the cost model, bench flags...
  tag=$1; general=$2; shift 2
  echo "== $tag: rocprofv3 passes"; bash tools/profile_round.sh $tag --no-reference-kernel "$@" > gpurun_out/profile_round_$tag.log 2>&1; tail -2 gpurun_out/profile_round_$tag.log
  echo "== $tag: scheduler statistics"; python bench.py --steps 3 --warmup 1 --scheduler-stats --no-cpu-baseline --no-boundary --no-reference-kernel "$@" > gpurun_out/${tag}_bench_scheduler_stats.json 2> gpurun_out/${tag}_bench_scheduler_stats.err
  python tools/valu_cost_model.py gpurun_out/${tag}_bench_scheduler_stats.json gpurun_out/pmc_$tag.json $general > gpurun_out/${tag}_valu_cost_model.json 2> gpurun_out/${tag}_valu_cost_model.err || tail -3 gpurun_out/${tag}_valu_cost_model.err
  # (bench.py finds the passes under profiles/: put this run's there for the line below - on the box only; the committed copies
  # are made from gpurun_out/ afterwards)
  scene=$(python -c "import json;print(json.load(open('gpurun_out/pmc_$tag.json'))['_config']['scene'])")
  cp gpurun_out/pmc_$tag.json profiles/r04_pmc_${scene}_default.json; cp gpurun_out/${tag}_valu_cost_model.json profiles/r04_valu_cost_model_${scene}_default.json
  echo "== $tag: bench line"; python bench.py --steps 6 --warmup 1 "$@" > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err || tail -3 gpurun_out/${tag}_bench.err
  python - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench.json').read().strip().splitlines()[-1])
r=d['roofline']; print('$tag', round(d['value'],1), d['unit'], 'vs reference kernel', d['vs_baseline'], '| bound', r['bound'], r['frac'], {k:(round(v['frac'],3) if 'frac' in v else None) for k,v in r.get('binding',{}).items() if isinstance(v,dict)}, '| cpu', d.get('cpu_baseline',{}).get('value'))
PY
}
case $part in
  tris1m)
    one_scene r04_tris1m ""
    echo "== strict arithmetic"; python bench.py --steps 6 --warmup 1 --arithmetic strict --no-cpu-baseline --no-boundary > gpurun_out/r04_tris1m_bench_strict.json 2> gpurun_out/r04_tris1m_bench_strict.err;;
  mayalike)
    one_scene r04_mayalike "--general --narrow" --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25;;
  cornell)
    one_scene r04_cornell "" --scene cornell --depth 8;;
  configs)
    bash tools/bench_configs.sh config0 config1 matmix tris4m
    echo "== north star at full sizes"; python tools/north_star_full_size.py tris1m cornell mayalike 2>&1 | grep -v "^{" | tail -8;;
esac
