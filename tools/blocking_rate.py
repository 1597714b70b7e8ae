#!/usr/bin/env python3
"""Mpaths/s of a caller that asks for ONE image per call and waits for it (the reference's loop, OpenCL.cpp:76-107), with and
without rendering ahead, and of 32 images per call.  usage: tools/blocking_rate.py [scene [W H depth]]   (PTMI_LIBRARY: a variant;
BLOCKING_RATE_DEVICES=0,1,...: an in-library multi-device context, where each device renders ahead of its own calls)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend
scene = sys.argv[1] if len(sys.argv) > 1 else "tris1m"
W, H, D = (int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080, 10)
sc = pt.bvh_create(pt.scenes.build(scene, W, H))
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
out = {}
DEVICES = [int(x) for x in os.environ["BLOCKING_RATE_DEVICES"].split(",")] if os.environ.get("BLOCKING_RATE_DEVICES") else None
DEPTH = os.environ.get("BLOCKING_RATE_DEPTH", "2")  # launches kept in flight ahead of the caller
for ahead, calls in ((DEPTH, "4"), (DEPTH, "2"), (DEPTH, "1"), ("0", "1")):
    os.environ["PTMI_RENDER_AHEAD"] = ahead
    os.environ["PTMI_RENDER_AHEAD_CALLS"] = calls
    be = pt.Backend().setup_context(W, H, D, sc.lightsSize, flags=backend.FLAG_DEFAULT_ARITHMETIC, devices=DEVICES)
    be.initialize_memory(sc)
    for k in range(100, 116):
        be.render(k, 1); be.synchronize()
    # an idle device on both sides of the timed region: what ran ahead before it and what is left running after it cancel
    hip.hipDeviceSynchronize()
    n = 32
    t0 = time.perf_counter()
    for k in range(116, 116 + n):
        be.render(k, 1); be.synchronize()
    hip.hipDeviceSynchronize()
    dt = time.perf_counter() - t0
    out[(ahead, calls)] = W * H * n / dt / 1e6
    if ahead == "0":
        for per_call in (4, 8):  # (what launches of that size reach when nothing waits in between)
            be.render(0, per_call); be.synchronize()
            t0 = time.perf_counter()
            for s in range(96 // per_call):
                be.render(300 + per_call * s, per_call)
            be.synchronize()
            out[str(per_call)] = W * H * 96 / (time.perf_counter() - t0) / 1e6
        be.render(0, 32); be.synchronize()
        t0 = time.perf_counter()
        for s in range(3):
            be.render(200 + 32 * s, 32)
        be.synchronize()
        out["32"] = W * H * 96 / (time.perf_counter() - t0) / 1e6
    be.release()
print(f"{scene}: one image per call, blocking, Mpaths/s: launches ahead for 4 calls {out[(DEPTH, '4')]:.1f}, for 2 calls {out[(DEPTH, '2')]:.1f}, "
      f"for 1 call {out[(DEPTH, '1')]:.1f}, nothing ahead {out[('0', '1')]:.1f}; not blocking: 4 images per call {out['4']:.1f}, 8 {out['8']:.1f}, 32 {out['32']:.1f}")
