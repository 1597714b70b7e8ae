#!/bin/bash
# round 4, tenth GPU call: 128-register builds (four waves per SIMD) where the tree depth already limits a CU to four workgroups
A="--no-reference-kernel"
echo "== mayalike 4K (general, depth 23: 4 workgroups per CU by LDS)"; STEPS=2 ROUNDS=2 BENCH_ARGS="$A --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25" bash tools/run_variants.sh base w4
echo "== tris4m (plain, depth 24)"; STEPS=2 ROUNDS=2 BENCH_ARGS="$A --scene tris4m" bash tools/run_variants.sh base w4
echo "== tris1m (plain, depth 22: 5 workgroups) for reference"; STEPS=3 ROUNDS=1 BENCH_ARGS="$A" bash tools/run_variants.sh base w4
echo "== nansafe rate"; timeout -k 10 900 python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json; echo "rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_nansafe_rate_partial.json"))
for k,e in d["scenes"].items():
    h=e["hostile_wavefront_nansafe"]
    print(k, "nansafe/clean", round(e["nansafe_over_clean"],3), "at the median launch", round(e["nansafe_over_clean_at_the_median_launch"],3), "launch seconds", h["seconds_per_launch"], "retraced", h["paths_retraced"])
PY
