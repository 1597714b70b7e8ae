#!/bin/bash
# round 4, first GPU call: fixture of the new parity case, the full-size reference comparisons, baseline bench lines
mkdir -p gpurun_out/golden
echo "== fixture"; python tests/golden/make_reference_fixtures.py gpurun_out/golden mayalike_s_96x96_d8 > gpurun_out/r04_fixture.log 2>&1 || { tail -5 gpurun_out/r04_fixture.log; exit 1; }
tail -3 gpurun_out/r04_fixture.log
echo "== full-size parity"; timeout -k 10 900 python -m pytest tests/test_reference_default_gpu.py -m gpu -x -q -k "full_size" > gpurun_out/r04_fullsize.log 2>&1; echo "rc $?"; tail -5 gpurun_out/r04_fullsize.log
echo "== bench mayalike"; timeout -k 10 900 python bench.py --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25 --steps 2 --warmup 1 > gpurun_out/r04_bench_mayalike_base.json 2> gpurun_out/r04_bench_mayalike_base.err; echo "rc $?"; tail -2 gpurun_out/r04_bench_mayalike_base.err
python - <<'PY'
import json
for n in ("mayalike_base",):
    try:
        d=json.load(open(f"gpurun_out/r04_bench_{n}.json"))
        print(n, round(d["value"],1), d["unit"], "Mpaths/s", round(d["Mpaths/s"],1), "vs_ref", d.get("vs_baseline"), "ref", d.get("reference_kernel",{}).get("Mpaths/s"), "eq-launch", d.get("reference_kernel",{}).get("ratio_at_equal_launch_counts"))
    except Exception as e: print(n, "failed", e)
PY
echo "== bench tris1m"; timeout -k 10 600 python bench.py --steps 4 --warmup 1 > gpurun_out/r04_bench_tris1m_base.json 2> gpurun_out/r04_bench_tris1m_base.err; echo "rc $?"
echo "== bench tris1m megakernel"; timeout -k 10 600 python bench.py --kernel megakernel --steps 2 --warmup 1 --no-boundary --no-cpu-baseline > gpurun_out/r04_bench_tris1m_megakernel.json 2> gpurun_out/r04_bench_tris1m_megakernel.err; echo "rc $?"
echo "== bench mayalike scheduler stats"; timeout -k 10 600 python bench.py --scene mayalike --width 3840 --height 2160 --depth 16 --spp-per-step 25 --steps 2 --warmup 1 --scheduler-stats --no-boundary --no-cpu-baseline --no-reference-kernel > gpurun_out/r04_bench_mayalike_sched.json 2> gpurun_out/r04_bench_mayalike_sched.err; echo "rc $?"
python - <<'PY'
import json
for n in ("tris1m_base","tris1m_megakernel","mayalike_sched"):
    try:
        d=json.load(open(f"gpurun_out/r04_bench_{n}.json"))
        print(n, round(d["value"],1), d["unit"], "Mpaths/s", round(d["Mpaths/s"],1), "vs_ref", d.get("vs_baseline"), d.get("wave_scheduler"))
    except Exception as e: print(n, "failed", e)
PY
