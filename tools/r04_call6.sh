#!/bin/bash
# round 4, sixth GPU call: top-of-stack in a register, per instantiation kind, on six workloads (same box)
A="--no-reference-kernel"
echo "== tris1m (plain)"; STEPS=3 ROUNDS=2 BENCH_ARGS="$A" bash tools/run_variants.sh
echo "== tris1m (general shading forced)"; PTMI_GENERIC_SHADING=1 STEPS=3 ROUNDS=2 BENCH_ARGS="$A" bash tools/run_variants.sh
echo "== cornell 1080p (plain)"; STEPS=6 ROUNDS=2 BENCH_ARGS="$A --scene cornell --depth 8" bash tools/run_variants.sh
echo "== matmix 4K (general)"; STEPS=3 ROUNDS=2 BENCH_ARGS="$A --scene matmix --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh
echo "== mayalike 4K (general, depth 23)"; STEPS=2 ROUNDS=2 BENCH_ARGS="$A --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh
echo "== tris4m (plain, depth 24)"; STEPS=2 ROUNDS=1 BENCH_ARGS="$A --scene tris4m" bash tools/run_variants.sh
echo "== config0 bench (readback path warmed)"; python bench.py --scene cornell --width 512 --height 512 --depth 4 --steps 2 --warmup 1 --cpu-spp 64 --cpu-rows 512 > gpurun_out/r04_bench_config0.json 2> gpurun_out/r04_bench_config0.err; python -c "
import json; d=json.load(open('gpurun_out/r04_bench_config0.json')); r=d['reference_kernel']; print('config0', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'vs_ref', d['vs_baseline'], 'blocking', r['ratio_at_equal_launch_counts'], r['ratio_at_equal_launch_counts_without_rendering_ahead'])"
