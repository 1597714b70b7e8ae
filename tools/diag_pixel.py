"""One pixel, sample by sample: this integrator vs the reference's default build (oracle/_ref), 1 spp per launch, and the
CPU oracle's bounce-by-bounce trace of the first samples that differ.  Diagnostic; runs on the GPU box.

usage: python tools/diag_pixel.py case x y [n_iterations]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, Backend  # noqa: E402

case, px, py = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
n_it = int(sys.argv[4]) if len(sys.argv) > 4 else 64
name, sampler, w, h, d = cases.CASES[case]
sc = bvh_create(scenes.build(name, w, h))
be = Backend().setup_context(w, h, d, sc.lightsSize, sampler)
be.initialize_memory(sc)
shown = 0
for it in range(n_it):
    r, _, _, _ = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it)
    be.clear()
    be.render(it, 1)
    g, _ = be.read_image()
    a, b = g[py, px, :3], r[py, px, :3]
    n_diff = int((np.abs(g[..., :3] - r[..., :3]).max(-1) > 1e-3 * np.maximum(np.abs(r[..., :3]).max(-1), 1e-6)).sum())
    flag = "" if np.allclose(a, b, rtol=1e-4, atol=1e-7) else "   <-- differs"
    print(f"it {it:3d}: ours {a} ref {b} (pixels differing in this image: {n_diff}){flag}")
    if flag and shown < 3:
        shown += 1
        bounces, rad = O.oracle_trace(sc, w, h, d, px, py, it, sampler)
        print(f"    oracle radiance {rad[:3]}")
        for k, bo in enumerate(bounces):
            mat = sc.materiaux[bo.material_id]
            print(f"    bounce {k}: tri {bo.triangle_id} mat {bo.material_id} type {int(mat['type'])} s {bo.s:.6f} t {bo.t:.6f} point {np.array(bo.point[:]).round(5)} "
                  f"Ns {np.array(bo.ns[:]).round(5)} out {np.array(bo.out_dir[:]).round(6)} transfer {np.array(bo.transfer[:3]).round(5)} rad {np.array(bo.radiance[:3]).round(5)} seed {bo.seed_after}")
be.release()
