"""Diagnostic (GPU box): per-sample comparison of the reference kernel vs the HIP integrator vs the oracle at 1 spp."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa
import oracle_ffi as O  # noqa
from opencl_pathtracer_amd import scenes, bvh_create, render_scene  # noqa

for case in sys.argv[1:] or ["cornell_64x48_d4", "matmix_96x96_d8"]:
    name, sampler, w, h, d = cases.CASES[case]
    sc = bvh_create(scenes.build(name, w, h))
    for it in (0, 1):
        r, rn, (rd, rb, rt), _ = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it)
        g, gn, (gd, gb, gt), _ = render_scene(sc, w, h, d, 1, first_iteration=it, sampler=sampler)
        a, b = r[..., :3].astype(np.float64), g[..., :3].astype(np.float64)
        rel = np.abs(a - b).max(-1) / np.maximum(np.maximum(np.abs(a), np.abs(b)).max(-1), 1e-12)
        hist = np.histogram(rel[rel > 0], bins=[0, 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 10])[0]
        print(f"{case} it{it}: pixels equal {int((rel == 0).sum())}/{rel.size}; rel-diff histogram "
              f"(<1e-7,<1e-6,<1e-5,<1e-4,<1e-3,<1e-2,<1e-1,rest) {hist.tolist()}; bbx-hist L1 "
              f"{int(np.abs(rb.astype(np.int64) - gb.astype(np.int64)).sum())} tri-hist L1 "
              f"{int(np.abs(rt.astype(np.int64) - gt.astype(np.int64)).sum())} depth L1 "
              f"{int(np.abs(rd.astype(np.int64) - gd.astype(np.int64)).sum())}")
        worst = np.argsort(rel.ravel())[::-1][:6]
        for k in worst:
            y, x = divmod(int(k), w)
            tr, _ = O.oracle_trace(sc, w, h, d, x, y, it, sampler)
            desc = [(t.triangle_id, t.material_id, round(t.s, 4), round(t.t, 4)) for t in tr]
            print(f"   px({x},{y}) ref {r[y, x, :3]} hip {g[y, x, :3]} rel {rel[y, x]:.2e} bounces(tri,mat,s,t) {desc}")
