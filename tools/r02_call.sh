python -m pytest tests -m gpu -x -q > gpurun_out/r02_t2_pytest.log 2>&1; tail -15 gpurun_out/r02_t2_pytest.log
bash tools/run_variants.sh r01kernel hit8 hit4 r01kernel hit8 hit4 2>&1 | tee gpurun_out/r02_variants_a.log
python tools/diag_features.py 256 gpurun_out/r02_features_256.json 2>&1 | tee gpurun_out/r02_features_256.log
