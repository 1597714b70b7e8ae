#!/bin/bash
# Round-2 evidence, one box: GPU tests, the default bench line, the reference kernel timed beside it, BASELINE's configs,
# rocprofv3 kernel stats + PMC passes for the headline workload and the 4M-triangle one.  Run from the repo root ON THE GPU BOX;
# copy the summaries from gpurun_out/ into profiles/ afterwards (tools/r02_collect.sh).
python -m pytest tests -m gpu -q -x > gpurun_out/r02_final_pytest.log 2>&1; tail -3 gpurun_out/r02_final_pytest.log
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err; cut -c1-400 gpurun_out/r02_final_bench.json
python tools/time_reference_kernel.py tris1m_1920x1080_d10 2 > gpurun_out/r02_final_ref_tris1m.json 2>&1; cat gpurun_out/r02_final_ref_tris1m.json
python tools/time_reference_kernel.py cornell_1920x1080_d8 4 > gpurun_out/r02_final_ref_cornell.json 2>&1; cat gpurun_out/r02_final_ref_cornell.json
bash tools/bench_configs.sh
bash tools/profile_round.sh r02_tris1m | tail -3
bash tools/profile_round.sh r02_tris4m --scene tris4m | tail -3
python bench.py --no-cpu-baseline --no-boundary --scheduler-stats > gpurun_out/r02_final_bench_sched.json 2>> gpurun_out/r02_final_bench.err
