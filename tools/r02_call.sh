STEPS=3 BENCH_ARGS="--spp-per-step 32" bash tools/run_variants.sh it64 2>&1 | tee gpurun_out/r02_variants_d.log
STEPS=2 BENCH_ARGS="--spp-per-step 64" bash tools/run_variants.sh it64 2>&1 | tee -a gpurun_out/r02_variants_d.log
bash tools/run_variants.sh shortdiv 2>&1 | tee -a gpurun_out/r02_variants_d.log
