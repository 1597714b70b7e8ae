#!/usr/bin/env python3
"""Which PIXELS of a scene differ between the integrator (both kernels), the reference kernel and the oracle at a given size?
(GPU box.)  usage: tools/diag_fuzz_pixels.py SCENE CODE_OBJECT W H DEPTH [SPP]   (default arithmetic)
Prints the differing pixels per iteration and the oracle's trace of the first few."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, render_scene  # noqa: E402


def main():
    name, case, w, h, d = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    spp = int(sys.argv[6]) if len(sys.argv) > 6 else 2
    warnings.simplefilter("ignore")
    sc = bvh_create(scenes.build(name, w, h))
    for it in range(spp):
        r = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it)
        g = render_scene(sc, w, h, d, 1, first_iteration=it, flags=16)
        m = render_scene(sc, w, h, d, 1, first_iteration=it, flags=16 | 2)
        bad = np.argwhere((r[0].view(np.uint32) != g[0].view(np.uint32)).any(-1))
        badm = np.argwhere((r[0].view(np.uint32) != m[0].view(np.uint32)).any(-1))
        print(f"iteration {it}: wavefront differs from the reference in {len(bad)} pixels, one-path-per-lane in {len(badm)}; depth histograms "
              f"ref {r[2][0].tolist()} wavefront {g[2][0].tolist()}", flush=True)
        for y, x in bad[:6].tolist():
            b, rad = O.oracle_trace(sc, w, h, d, x, y, it, default_arithmetic=True)
            print(f"  pixel ({x}, {y}): ref {r[0][y, x].tolist()} wavefront {g[0][y, x].tolist()} one-path-per-lane {m[0][y, x].tolist()} oracle {rad.tolist()} ({len(b)} bounces)")
            for i, q in enumerate(b[:4]):
                tr = sc.triangulation[q.triangle_id]
                print(f"    bounce {i}: triangle {q.triangle_id} material {q.material_id} (type {int(sc.materiaux['type'][q.material_id])}) s {q.s:.6g} t {q.t:.6g} point {list(q.point)}")
                print(f"      S1 {tr['S1'].tolist()} S2 {tr['S2'].tolist()} S3 {tr['S3'].tolist()} N {tr['N'].tolist()}")
                print(f"      ns {list(q.ns)} out {list(q.out_dir)} transfer {list(q.transfer)} boxes {q.n_bbx} triangles {q.n_tri}")


if __name__ == "__main__":
    main()
