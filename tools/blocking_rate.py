#!/usr/bin/env python3
"""Mpaths/s of a caller that asks for ONE image per call and waits for it (the reference's loop, OpenCL.cpp:76-107), with and
without rendering ahead, and of 32 images per call.  usage: tools/blocking_rate.py [scene [W H depth]]   (PTMI_LIBRARY: a variant)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend
scene = sys.argv[1] if len(sys.argv) > 1 else "tris1m"
W, H, D = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 10)
sc = pt.bvh_create(pt.scenes.build(scene, W, H))
out = []
for ahead in ("2", "0"):
    os.environ["PTMI_RENDER_AHEAD"] = ahead
    be = pt.Backend().setup_context(W, H, D, sc.lightsSize, flags=backend.FLAG_DEFAULT_ARITHMETIC)
    be.initialize_memory(sc)
    for k in range(100, 103):
        be.render(k, 1); be.synchronize()
    n = 24
    t0 = time.perf_counter()
    for k in range(103, 103 + n):
        be.render(k, 1); be.synchronize()
    dt = time.perf_counter() - t0
    out.append(W * H * n / dt / 1e6)
    if ahead == "0":
        be.render(0, 32); be.synchronize()
        t0 = time.perf_counter()
        for s in range(3):
            be.render(200 + 32 * s, 32)
        be.synchronize()
        out.append(W * H * 96 / (time.perf_counter() - t0) / 1e6)
    be.release()
print(f"{scene}: one image per call, blocking: {out[0]:.1f} Mpaths/s rendering ahead, {out[1]:.1f} without; 32 per call {out[2]:.1f}")
