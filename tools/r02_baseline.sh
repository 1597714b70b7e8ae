set -e
python -m pytest tests -m gpu -x -q > gpurun_out/r02_base_pytest.log 2>&1
python bench.py --steps 6 --warmup 2 > gpurun_out/r02_base_bench.json 2> gpurun_out/r02_base_bench.err
python tools/time_reference_kernel.py tris1m_1920x1080_d10 2 > gpurun_out/r02_base_ref_tris1m.json 2>&1
python tools/time_reference_kernel.py cornell_1920x1080_d8 4 > gpurun_out/r02_base_ref_cornell.json 2>&1
tail -2 gpurun_out/r02_base_pytest.log; cat gpurun_out/r02_base_bench.json gpurun_out/r02_base_ref_tris1m.json gpurun_out/r02_base_ref_cornell.json
