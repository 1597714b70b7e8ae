#!/usr/bin/env python3
"""Which (pixel, iteration) of the configs[4] stand-in differs between the wavefront kernel, the one-path-per-lane kernel and the
reference kernel?  (GPU box.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import oracle_ffi as O
import opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend, scenes
W, H, D = 3840, 2160, 16
DA = backend.FLAG_DEFAULT_ARITHMETIC
sc = pt.bvh_create(scenes.build("mayalike", W, H))
case = "mayalike_3840x2160_d16"
found = 0
for it in range(0, 64):
    ref = O.ref_gpu_render(case, sc, W, H, D, 1, first_iteration=it)
    wf = pt.render_scene(sc, W, H, D, 1, first_iteration=it, flags=DA)
    bad = np.argwhere((wf[0].view(np.uint32) != ref[0].view(np.uint32)).any(-1))
    same_hist = all(np.array_equal(a, b) for a, b in zip(wf[2], ref[2]))
    print("iteration", it, "differing pixels", len(bad), "histograms equal", same_hist, flush=True)
    if len(bad):
        mk = pt.render_scene(sc, W, H, D, 1, first_iteration=it, flags=DA | backend.FLAG_MEGAKERNEL)
        for (y, x) in bad[:6]:
            print("   pixel", int(x), int(y), "wavefront", wf[0][y, x], "reference", ref[0][y, x], "one-path-per-lane", mk[0][y, x], flush=True)
        be = pt.Backend().setup_context(W, H, D, sc.lightsSize, flags=DA); be.initialize_memory(sc); be.render(it, 1); be.synchronize()
        print("   paths retraced in this iteration:", be.scheduler_stats()["paths_retraced"]); be.release()
        os.environ["PTMI_WALK_NAN_RAYS"] = "1"
        wf2 = pt.render_scene(sc, W, H, D, 1, first_iteration=it, flags=DA)
        del os.environ["PTMI_WALK_NAN_RAYS"]
        print("   with PTMI_WALK_NAN_RAYS: differing pixels", int((wf2[0].view(np.uint32) != ref[0].view(np.uint32)).any(-1).sum()))
        found += 1
        if found >= 3:
            break
