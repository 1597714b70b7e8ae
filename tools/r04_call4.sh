#!/bin/bash
# round 4, fourth GPU call: render-ahead tests + the whole GPU suite, L1 ceiling microbench, NANSAFE rates, bench lines
echo "== render-ahead tests"; timeout -k 10 600 python -m pytest tests/test_render_ahead_gpu.py -m gpu -x -q 2>&1 | tail -15
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; echo "rc $?"; tail -6 gpurun_out/r04_gputests.log
echo "== l1 ceiling"; timeout -k 10 600 bash tools/l1_ceiling.sh 2>&1 | tail -22
echo "== bench tris1m"; timeout -k 10 600 python bench.py --steps 4 --warmup 1 > gpurun_out/r04_bench_tris1m_ahead.json 2> gpurun_out/r04_bench_tris1m_ahead.err; echo "rc $?"
echo "== bench config0"; timeout -k 10 600 python bench.py --scene cornell --width 512 --height 512 --depth 4 --steps 2 --warmup 1 --cpu-spp 64 --cpu-rows 512 > gpurun_out/r04_bench_config0.json 2> gpurun_out/r04_bench_config0.err; echo "rc $?"
python - <<'PY'
import json
for n in ("tris1m_ahead","config0"):
    try:
        d=json.load(open(f"gpurun_out/r04_bench_{n}.json")); r=d.get("reference_kernel",{})
        print(n, round(d["value"],1), d["unit"], "Mpaths/s", round(d["Mpaths/s"],1), "vs_ref", d.get("vs_baseline"), "| blocking caller:", r.get("ratio_at_equal_launch_counts"), "without ahead:", r.get("ratio_at_equal_launch_counts_without_rendering_ahead"), "| boundary", {k:round(v,1) for k,v in d.get("boundary",{}).items() if k.endswith("Msamples/s")})
    except Exception as e: print(n, "failed", e)
PY
echo "== nansafe rate"; timeout -k 10 900 python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json; echo "rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_nansafe_rate_partial.json"))
for k,e in d["scenes"].items():
    print(k, {n:(round(v["Msamples/s"],1), v["paths_retraced"]) for n,v in e.items() if isinstance(v,dict)}, "nansafe/clean", round(e["nansafe_over_clean"],3), "nansafe/megakernel", round(e["nansafe_over_one_path_per_lane"],2), "retraced share", round(e["share_of_paths_retraced"],5))
PY
