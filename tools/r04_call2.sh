#!/bin/bash
# round 4, second GPU call: the whole GPU suite on the new kernel source, the NANSAFE rates
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; echo "rc $?"; tail -8 gpurun_out/r04_gputests.log
echo "== nansafe rate"; timeout -k 10 600 python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json 2> gpurun_out/r04_nansafe_rate.err; echo "rc $?"; tail -3 gpurun_out/r04_nansafe_rate.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_nansafe_rate.json"))
for k,e in d["scenes"].items():
    print(k, {n:(round(v["Msamples/s"],1), v["paths_retraced"]) for n,v in e.items() if isinstance(v,dict)}, round(e["nansafe_over_clean"],3), round(e["nansafe_over_one_path_per_lane"],2))
PY
