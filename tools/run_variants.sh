#!/bin/bash
# Bench every opencl_pathtracer_amd/lib/variants/libptmi_*.so (tools/build_variants.sh) on this box, same command each.
# usage: tools/run_variants.sh [name ...]      env: STEPS (3), BENCH_ARGS (extra bench.py flags), ROUNDS (1: passes over the list)
mkdir -p gpurun_out
names="$@"; [ -z "$names" ] && names=$(ls opencl_pathtracer_amd/lib/variants/libptmi_*.so | sed 's/.*libptmi_//; s/\.so$//')
for round in $(seq ${ROUNDS:-1}); do
for name in $names; do
  PTMI_LIBRARY=$PWD/opencl_pathtracer_amd/lib/variants/libptmi_$name.so timeout -k 10 120 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline --no-boundary ${BENCH_ARGS} 2> gpurun_out/variant_$name.err | grep -E "^\{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step')" || { echo "$name: RUN FAILED"; tail -3 gpurun_out/variant_$name.err; }
done
done
