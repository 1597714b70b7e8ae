#!/bin/bash
# Build variants of libptmi.so with different -D tuning macros HERE (hipcc cross-compiles without a GPU); the .so files
# under opencl_pathtracer_amd/lib/variants/ travel to the GPU box with the snapshot, where tools/run_variants.sh benches
# them one after the other on the same box.  Each variant holds both arithmetic modes, like the product library (Makefile).
# usage: tools/build_variants.sh "name1:-DA=1 -DB=2" "name2:-DA=3" ...       (SRC=<dir> builds another source tree)
set -e
SRC=${SRC:-.}
CSRC=$SRC/opencl_pathtracer_amd/csrc
OUT=opencl_pathtracer_amd/lib/variants
mkdir -p $OUT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I$SRC/include -I$CSRC"
build_one() {
  name="$1"; defs="$2"; obj=$OUT/obj_$name; mkdir -p $obj
  { for f in kernels kernel_wavefront; do
      /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/$f.hip -o $obj/$f.o &
      /opt/rocm/bin/hipcc $FLAGS $defs -DPTMI_DEFAULT_ARITHMETIC=1 -c $CSRC/$f.hip -o $obj/${f}_da.o &
    done
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/display.hip -o $obj/display.o &
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/ptmi_api.cpp -o $obj/ptmi_api.o &
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/bvh_build.cpp -o $obj/bvh_build.o &
    /opt/rocm/bin/hipcc $FLAGS $defs -c $CSRC/scene_layout.cpp -o $obj/scene_layout.o &
    wait; } 2> $OUT/build_$name.log
  /opt/rocm/bin/hipcc $FLAGS -shared $obj/*.o -o $OUT/libptmi_$name.so 2>> $OUT/build_$name.log && echo "built $name" || { echo "$name: BUILD FAILED"; tail -5 $OUT/build_$name.log; }
  rm -rf $obj
}
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"; [ "$defs" = "$spec" ] && defs=""
  build_one "$name" "$defs"
done
