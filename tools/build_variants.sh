#!/bin/bash
# Build variants of libptmi.so with different -D tuning macros HERE (hipcc cross-compiles without a GPU); the .so files
# under opencl_pathtracer_amd/lib/variants/ travel to the GPU box with the snapshot, where tools/run_variants.sh benches
# them one after the other on the same box.
# usage: tools/build_variants.sh "name1:-DA=1 -DB=2" "name2:-DA=3" ...       (SRC=<dir> builds another source tree)
set -e
SRC=${SRC:-.}
CSRC=$SRC/opencl_pathtracer_amd/csrc
OUT=opencl_pathtracer_amd/lib/variants
mkdir -p $OUT
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"; [ "$defs" = "$spec" ] && defs=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC/include -I$CSRC $defs \
      -shared $CSRC/kernels.hip $CSRC/kernel_wavefront.hip $CSRC/display.hip $CSRC/ptmi_api.cpp $CSRC/bvh_build.cpp -o $OUT/libptmi_$name.so \
      2> $OUT/build_$name.log && echo "built $name" || { echo "$name: BUILD FAILED"; tail -5 $OUT/build_$name.log; }
done
