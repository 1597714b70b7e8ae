#!/bin/bash
# round 4, eighth GPU call: the whole GPU suite on the current source, then bench lines
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; echo "rc $?"; tail -6 gpurun_out/r04_gputests.log
echo "== smoke"; python __graft_entry__.py smoke 2>&1 | tail -3
echo "== nansafe rate"; timeout -k 10 900 python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json; echo "rc $?"
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_nansafe_rate_partial.json"))
for k,e in d["scenes"].items():
    print(k, {n:(round(v["Msamples/s"],1), v["paths_retraced"]) for n,v in e.items() if isinstance(v,dict) and "Msamples/s" in v}, "nansafe/clean", round(e["nansafe_over_clean"],3), "nansafe/megakernel", round(e["nansafe_over_one_path_per_lane"],2), "vs reference kernel", e.get("nansafe_over_reference_kernel"), e.get("reference_kernel_on_the_hostile_scene"))
PY
