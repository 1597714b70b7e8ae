#!/usr/bin/env python3
"""Where does the wall time of a SHORT bench region go (Cornell box 512x512: 3.9 ms launches)?  Times, on the GPU box, the same
sequence bench.py times - K steps of 32 iterations, wait, one readback - for several K, and its pieces."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import opencl_pathtracer_amd as pt  # noqa: E402
from opencl_pathtracer_amd import backend  # noqa: E402

W = H = 512
D, B = 4, 32
sc = pt.bvh_create(pt.scenes.build("cornell", W, H))
be = pt.Backend().setup_context(W, H, D, sc.lightsSize, pt.structs.JITTERED, flags=backend.FLAG_DEFAULT_ARITHMETIC)
be.initialize_memory(sc)
out = (np.empty((H, W, 4), np.float32), np.empty((H, W), np.float32))
be.pin_host_buffer(out[0]); be.pin_host_buffer(out[1])
be.render(0, B); be.synchronize()
for K in (1, 2, 4, 8, 16, 32):
    for rep in range(2):
        be.synchronize(); be.kernel_time()
        t0 = time.perf_counter()
        for s in range(K):
            be.render(1000 + s * B, B)
        t1 = time.perf_counter()
        be.synchronize()
        t2 = time.perf_counter()
        be.read_image(out=out)
        t3 = time.perf_counter()
        ms, n = be.kernel_time()
        print(f"K={K:2d} rep {rep}: enqueue {1e3*(t1-t0):6.2f} ms  wait {1e3*(t2-t1):6.2f} ms  read {1e3*(t3-t2):5.2f} ms  total/K {1e3*(t3-t0)/K:6.2f} ms   "
              f"event time/launch {ms/max(n,1):5.2f} ms ({n} launches)", flush=True)
be.release()
