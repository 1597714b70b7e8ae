#!/bin/bash
# round 4, seventh GPU call: NaN-ray probe (RANDOM sampler re-trace), parameter sweeps on the configs[4] stand-in, w-trivial leaf pass
A="--no-reference-kernel"
echo "== NaN-ray probe"; timeout -k 10 300 python tools/diag_nan_rays.py 2>&1 | tail -14
echo "== mayalike 4K (general, depth 23)"; STEPS=2 ROUNDS=2 BENCH_ARGS="$A --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh base wd512 wd1024 wd1536 ll15 ll24
echo "== tris1m (plain)"; STEPS=3 ROUNDS=2 BENCH_ARGS="$A" bash tools/run_variants.sh base wtriv ll15 ll24
echo "== cornell 1080p (plain)"; STEPS=6 ROUNDS=2 BENCH_ARGS="$A --scene cornell --depth 8" bash tools/run_variants.sh base wtriv
echo "== matmix 4K (general)"; STEPS=3 ROUNDS=1 BENCH_ARGS="$A --scene matmix --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh base wd1024 ll15 ll24
