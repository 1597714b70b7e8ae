"""Times the REFERENCE kernel (oracle/_ref code object built from the unmodified .cl) on the GPU next to ours,
same scene, same iterations, and reports Msamples/s for both plus the per-channel RMS between the images.
Test/measurement infrastructure: uses oracle/_ref, never imported by the product or by bench.py.

usage: python tools/time_reference_kernel.py [case] [spp]     (default tris1m_1920x1080_d10, 2 spp)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import Backend, scenes, bvh_create, structs as S  # noqa: E402
from opencl_pathtracer_amd.backend import FLAG_NO_HISTOGRAMS  # noqa: E402

BIG = {"tris1m_1920x1080_d10": ("tris1m", S.JITTERED, 1920, 1080, 10), "cornell_1920x1080_d8": ("cornell", S.JITTERED, 1920, 1080, 8)}
case = sys.argv[1] if len(sys.argv) > 1 else "tris1m_1920x1080_d10"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2
name, sampler, w, h, d = {**cases.CASES, **BIG}[case]
sc = bvh_create(scenes.build(name, w, h))

r_color, r_count, (r_dep, _, _), ref_ms = O.ref_gpu_render(case, sc, w, h, d, spp)
seg_ref = int((np.minimum(np.arange(d + 1) + 1, d) * r_dep.astype(np.int64)).sum())  # upper bound: d+1 queries per path, at most D

be = Backend().setup_context(w, h, d, sc.lightsSize, sampler)
be.initialize_memory(sc)
be.render(0, spp)
be.synchronize()
be.clear()
be.kernel_time()
t0 = time.perf_counter()
be.render(0, spp)
be.synchronize()
wall = time.perf_counter() - t0
ms, _ = be.kernel_time()
color, count = be.read_image()
c = be.counters()
dep, _, _ = be.read_statistics()
# the integrator's own launch size: 32 iterations per launch (what OpenCL_RunKernel's loop gets through ptmi_render_snapshots
# and what bench.py times); the reference's rate does not depend on the count - it is one launch per iteration either way
be.clear()
be.kernel_time()
be.render(0, 32)
be.synchronize()
ms32, _ = be.kernel_time()
c32 = be.counters()
be.release()

rms = cases.rms_per_channel(color, count, r_color, r_count)
out = {"case": case, "spp": spp,
       "reference_kernel": {"launches": spp, "work_group": "8x8 (the reference's 1x1 would idle 63 of 64 lanes)",
                            "kernel_ms_total": ref_ms, "Mpaths/s": w * h * spp / ref_ms / 1e3,
                            "Msamples/s_upper_bound": seg_ref / ref_ms / 1e3},
       "this_integrator": {"kernel_ms_total": ms, "Mpaths/s": c["paths"] / ms / 1e3, "Msamples/s": c["segments"] / ms / 1e3},
       "speedup_paths": (c["paths"] / ms) / (w * h * spp / ref_ms),
       "this_integrator_32_iterations_per_launch": {"kernel_ms_total": ms32, "Mpaths/s": c32["paths"] / ms32 / 1e3,
                                                    "Msamples/s": c32["segments"] / ms32 / 1e3},
       "speedup_paths_32_iterations_per_launch": (c32["paths"] / ms32) / (w * h * spp / ref_ms),
       "rms_per_channel_vs_reference": rms.tolist(),
       "depth_histogram_L1": int(np.abs(dep.astype(np.int64) - r_dep.astype(np.int64)).sum()), "paths": int(dep.sum())}
print(json.dumps(out))
