#!/bin/bash
# round 4, third GPU call: the GPU suite on the reworked re-trace kernel, the NANSAFE rates, the L1 ceiling microbench
echo "== pytest -m gpu"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputests.log 2>&1; echo "rc $?"; tail -6 gpurun_out/r04_gputests.log
echo "== nansafe rate"; timeout -k 10 900 python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json 2> gpurun_out/r04_nansafe_rate.err; echo "rc $?"; tail -3 gpurun_out/r04_nansafe_rate.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_nansafe_rate.json"))
for k,e in d["scenes"].items():
    print(k, {n:(round(v["Msamples/s"],1), v["paths_retraced"]) for n,v in e.items() if isinstance(v,dict)}, "nansafe/clean", round(e["nansafe_over_clean"],3), "nansafe/megakernel", round(e["nansafe_over_one_path_per_lane"],2), "retraced share", round(e["share_of_paths_retraced"],5))
PY
echo "== l1 ceiling"; timeout -k 10 900 bash tools/l1_ceiling.sh 2>&1 | tail -20
