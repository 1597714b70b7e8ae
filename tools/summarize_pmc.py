#!/usr/bin/env python3
"""Per-launch means of the rocprofv3 --pmc counters of render_wavefront_kernel (input: the directory
tools/profile_round.sh wrote).  FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; bench.py converts
them to HBM bytes with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE x 2)."""
import csv, glob, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
for f in glob.glob(sys.argv[1] + "/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "render_wavefront_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
out = {}
for name, per_dispatch in acc.items():
    v = list(per_dispatch.values())[1:] or list(per_dispatch.values())  # drop the warm-up launch
    out[name] = {"per_launch_mean": sum(v) / len(v), "launches": len(v)}
extra = sys.argv[2] if len(sys.argv) > 2 else ""
m_spp = __import__("re").search(r"--spp-per-step[ =](\d+)", extra)
out["_spp_per_launch"] = int(m_spp.group(1)) if m_spp else 32  # bench.py's --spp-per-step (default 32) = ONE launch per step
# the kernel source these counters were taken on: bench.py quotes them only for that very source (bench.kernel_source_digest)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out["_kernel_source_digest"] = bench.kernel_source_digest()
# ... and the workload: bench.py's defaults unless the pass was given other flags
import re
cfg = {"scene": "tris1m", "width": 1920, "height": 1080, "depth": 10, "arithmetic": "default"}
for key in cfg:
    m = re.search(r"--%s[ =](\S+)" % key, extra)
    if m:
        cfg[key] = m.group(1) if key in ("scene", "arithmetic") else int(m.group(1))
out["_config"] = cfg
out["_note"] = ("rocprofv3 --pmc <one group per run> --kernel-trace -- python3 bench.py --steps 2 --warmup 1 "
                "--no-cpu-baseline --no-boundary " + extra + " (32 spp per launch, 1080p); per-launch means of "
                "render_wavefront_kernel; FETCH_SIZE/WRITE_SIZE in KB; GRBM_GUI_ACTIVE summed over the 8 XCDs")
print(json.dumps(out, indent=1))
