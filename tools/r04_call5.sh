#!/bin/bash
# round 4, fifth GPU call: A/B of kernel variants on one box, L1 ceiling (L2-resident cold set), short-region diagnostic
echo "== variants tris1m"; STEPS=3 ROUNDS=2 BENCH_ARGS="--no-reference-kernel" bash tools/run_variants.sh
echo "== variants cornell 1080p"; STEPS=6 ROUNDS=1 BENCH_ARGS="--no-reference-kernel --scene cornell --depth 8" bash tools/run_variants.sh
echo "== variants mayalike"; STEPS=2 ROUNDS=1 BENCH_ARGS="--no-reference-kernel --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh
echo "== short region"; timeout -k 10 300 python tools/diag_short_region.py 2>&1 | tail -14
echo "== l1 ceiling"; timeout -k 10 900 bash tools/l1_ceiling.sh 2>&1 | tail -24
