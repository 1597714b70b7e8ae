#!/bin/bash
# BASELINE.json's configs, one bench line each (run from the repo root ON THE GPU BOX; copy the JSON lines to profiles/).
#   configs[0] Cornell box 512x512, 64 spp, 4 bounces - the reference's CPU case: the CPU baseline renders it IN FULL
#   configs[1] Cornell box 1920x1080, 1024 spp, 8 bounces
#   configs[2] 1M random triangles 1920x1080, 256 spp (the headline workload; bench.py's default at 16 spp per step)
#   configs[4] stand-in (SURVEY 8d Config 5): scenes.maya_like - 1.09 M textured triangles, four 1024x1024 textures, 6 x 512x512 sky -
#              3840x2160, 16 bounces, 100 of its 2048 spp (the Maya asset does not exist); matmix = the toy material mix of rounds 1-3
#   tris4m     4M triangles = 427 MB of records, beyond the Infinity Cache (still not HBM-bound)
# usage: tools/bench_configs.sh [names...]
mkdir -p gpurun_out
run() { name=$1; shift; echo "== $name: bench.py $*"; python bench.py "$@" > gpurun_out/r04_bench_$name.json 2> gpurun_out/r04_bench_$name.err || tail -3 gpurun_out/r04_bench_$name.err;
        python -c "import json,sys; d=json.load(open('gpurun_out/r04_bench_$name.json')); print('$name', round(d['value'],1), d['unit'], '| roofline frac', round(d['roofline']['frac'],3), '| cpu', d.get('cpu_baseline',{}).get('value'), '| boundary', d.get('boundary',{}).get('per_image_vs_batched'))"; }
want="$@"; [ -z "$want" ] && want="config0 config1 config2 config4 matmix tris4m"
for n in $want; do case $n in
  config0) run config0_cornell_512_d4_64spp --scene cornell --width 512 --height 512 --depth 4 --steps 2 --warmup 1 --cpu-spp 64 --cpu-rows 512;;
  config1) run config1_cornell_1080p_d8_1024spp --scene cornell --depth 8 --steps 64 --warmup 0;;
  config2) run config2_tris1m_1080p_d10_256spp --steps 16 --warmup 0;;
  config4) run config4_standin_mayalike_4k_d16 --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25 --steps 4 --warmup 1;;
  matmix)  run matmix_4k_d16 --scene matmix --width 3840 --height 2160 --depth 16 --spp-per-step 25 --steps 4 --warmup 1;;
  tris4m)  run tris4m_1080p_d10 --scene tris4m --steps 4 --warmup 1;;
esac; done
