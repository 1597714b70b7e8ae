"""Where does a ONE-iteration launch spend its time?  Wall time of render(k, 1) + synchronize (the reference's per-image protocol
without the readback), the device time between the events around the launch's work, and - with rocprofv3 --kernel-trace --stats
around this script - the kernels' own durations.  usage: python tools/diag_short_launch.py [n_iterations_per_launch ...]"""
import sys
import time
sys.path[:0] = [".", "tests"]
import opencl_pathtracer_amd as pt

import os
W, H, D = 1920, 1080, int(os.environ.get("DEPTH", "10"))  # DEPTH=1: every path one segment long - a launch without a ragged end
sc = pt.bvh_create(pt.scenes.build("tris1m", W, H))
be = pt.Backend().setup_context(W, H, D, sc.lightsSize, flags=pt.backend.FLAG_DEFAULT_ARITHMETIC)
be.initialize_memory(sc)
be.render(1000, 4)
be.synchronize()
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 32]:
    be.kernel_time()
    reps = max(2, 32 // n)
    t0 = time.perf_counter()
    for k in range(reps):
        be.render(k * n, n)
        be.synchronize()
    wall = (time.perf_counter() - t0) / reps
    ms, launches = be.kernel_time()
    print(f"{n:2d} iterations per launch: wall {wall * 1e3:7.2f} ms per call, device {ms / launches:7.2f} ms per call, "
          f"{W * H * n / wall / 1e6:6.1f} Mpaths/s; per iteration {wall * 1e3 / n:6.2f} ms")
be.release()
