#!/bin/bash
# Build variants of libptmi.so with different -D tuning macros and bench each on the GPU box.
# usage: tools/sweep_variants.sh "name1:-DA=1 -DB=2" "name2:-DA=3" ...   (run from the repo root, on the GPU box)
set -e
CSRC=opencl_pathtracer_amd/csrc
mkdir -p gpurun_out/variants
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  out=gpurun_out/variants/libptmi_$name.so
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Iinclude -I$CSRC $defs \
      -shared $CSRC/kernels.hip $CSRC/kernel_wavefront.hip $CSRC/display.hip $CSRC/ptmi_api.cpp $CSRC/bvh_build.cpp -o $out 2> gpurun_out/variants/build_$name.log || { echo "$name: BUILD FAILED"; continue; }
  PTMI_LIBRARY=$PWD/$out timeout -k 10 90 python bench.py --steps ${STEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS} 2>&1 | grep -E "^\{" | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],1), 'ms/step frac', round(d['roofline']['frac'],3))" || echo "$name: RUN FAILED"
done
