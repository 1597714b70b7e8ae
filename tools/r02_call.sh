python -m pytest tests -m gpu -q -x > gpurun_out/r02_t8_pytest.log 2>&1; tail -3 gpurun_out/r02_t8_pytest.log
python bench.py --steps 6 --warmup 2 > gpurun_out/r02_bench_a.json 2> gpurun_out/r02_bench_a.err; cat gpurun_out/r02_bench_a.json
PTMI_SERIAL_LAUNCHES=1 python bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bench_a_serial.json 2> gpurun_out/r02_bench_a_serial.err; cat gpurun_out/r02_bench_a_serial.json
python tools/time_reference_kernel.py tris1m_1920x1080_d10 2 > gpurun_out/r02_ref_tris1m.json 2>&1; cat gpurun_out/r02_ref_tris1m.json
