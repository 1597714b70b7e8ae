#!/bin/bash
# round 4, ninth GPU call: the "one visit per hit" experiment (PTMI_WF_SWAP) - bit-exactness with the variant library, then A/B
A="--no-reference-kernel"
V=$PWD/opencl_pathtracer_amd/lib/variants
echo "== parity tests with the swap1 library"; PTMI_LIBRARY=$V/libptmi_swap1.so timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_reference_default_gpu.py tests/test_render_ahead_gpu.py -m gpu -q -x 2>&1 | tail -8
echo "== tris1m (plain)"; STEPS=3 ROUNDS=2 BENCH_ARGS="$A" bash tools/run_variants.sh swap0 swap1
echo "== cornell 1080p (plain)"; STEPS=6 ROUNDS=2 BENCH_ARGS="$A --scene cornell --depth 8" bash tools/run_variants.sh swap0 swap1
echo "== tris4m (plain)"; STEPS=2 ROUNDS=1 BENCH_ARGS="$A --scene tris4m" bash tools/run_variants.sh swap0 swap1
echo "== tris1m scheduler view (swap1 has no STATS build of the plain kernel; general for reference)"; 
echo "== which kernel takes the time with one zero-area triangle"
R=$PWD; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nansafe -o ns -- python3 $R/tools/diag_nansafe_kernels.py 2>&1 | tail -6
cd $R; find gpurun_out/prof_nansafe -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-5 {} | cut -c1-150 | head -8'
