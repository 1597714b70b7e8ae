bash tools/run_variants.sh head late3 head late3 2>&1 | tee gpurun_out/r02_variants_f.log
