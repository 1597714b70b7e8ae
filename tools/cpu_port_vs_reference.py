"""How fast is the CPU port (oracle/pt_oracle.c) next to the reference's own CPU path?

The reference's CPU path = its Kernel_Main on an OpenCL CPU device.  No OpenCL CPU runtime exists in this image, and building the
unmodified .cl for x86-64 needs an OpenCL builtin library the image lacks (writing a stand-in for it is ruled out), so that path
cannot be timed again here.  What exists is the survey's own measurement of it in this container class (SURVEY.md 6 / BASELINE.md 2,
throw-away probe of the survey stage: the unmodified .cl compiled for x86-64, one thread, Xeon @ 2.1 GHz):
    1M random triangles, 256x144, 2 spp, depth 10, 1 point light:  73 728 paths in 5.25 s = 0.0140 Mpaths/s/thread
This tool times the port on the same workload shape (same generator family, image size, depth, sampler, one thread; then all
cores) in this container and writes profiles/r03_cpu_reference_vs_port.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create  # noqa: E402

w, h, d, spp = 256, 144, 10, 2
sc = bvh_create(scenes.build("tris1m", w, h))
out = {"workload": f"tris1m {w}x{h}, {spp} spp, depth {d}, JITTERED, 1 point light ({len(sc.triangulation)} triangles)",
       "reference_cpu_path_survey_probe": {"Mpaths_per_s_per_thread": 73728 / 5.25 / 1e6, "paths": 73728, "seconds": 5.25,
                                           "box_tests_per_path": 1675, "triangle_tests_per_path": 490,
                                           "source": "SURVEY.md 6 [probe] / BASELINE.md 2: unmodified FullKernel.cl compiled for x86-64, 1 thread, this container class"}}
for arith in (False, True):
    for threads in (1, os.cpu_count()):
        O.oracle_render(sc, w, h, d, 1, n_threads=threads, default_arithmetic=arith)  # warm-up (tables, page faults)
        t0 = time.perf_counter()
        _, _, _, tot = O.oracle_render(sc, w, h, d, spp, n_threads=threads, default_arithmetic=arith)
        dt = time.perf_counter() - t0
        key = f"port_{'default' if arith else 'strict'}_arithmetic_{threads}_thread{'s' if threads > 1 else ''}"
        out[key] = {"Mpaths_per_s": tot["paths"] / dt / 1e6, "Msamples_per_s": tot["segments"] / dt / 1e6, "seconds": dt, "threads": threads,
                    "box_tests_per_path": tot["box_tests"] / tot["paths"], "triangle_tests_per_path": tot["triangle_tests"] / tot["paths"]}
        print(key, out[key], flush=True)
ref = out["reference_cpu_path_survey_probe"]["Mpaths_per_s_per_thread"]
out["port_over_reference_one_thread"] = {"strict": out["port_strict_arithmetic_1_thread"]["Mpaths_per_s"] / ref,
                                         "default": out["port_default_arithmetic_1_thread"]["Mpaths_per_s"] / ref}
out["host"] = {"cpus": os.cpu_count(), "model": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json"), "w"), indent=1)
print(json.dumps(out["port_over_reference_one_thread"]))
