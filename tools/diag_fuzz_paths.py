#!/usr/bin/env python3
"""Which PATHS of a fuzz scene count other numbers of box / triangle tests in the reference kernel than in the oracle?  (GPU box.)

usage: [STRICT=1] tools/diag_fuzz_paths.py SCENE_NAME [ITERATIONS]
One iteration at a time: the per-path histograms of the reference kernel and of the oracle differ in a few bins; the oracle's
per-path trace then names the pixel whose path holds the oracle's count, and prints its bounces."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create  # noqa: E402


def main():
    name = sys.argv[1]
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    strict = os.environ.get("STRICT") == "1"
    nl = int(name.rsplit("_l", 1)[1])
    case, w, h, d = ("feat_64x64_d8", 64, 64, 8) if nl == 1 else ("matmix_96x96_d8", 96, 96, 8)
    if os.environ.get("DEPTH") == "1":
        case, d = "feat_64x64_d1", 1
    warnings.simplefilter("ignore")
    sc = scenes.build(name, w, h)
    if os.environ.get("DROP_HOSTILE") == "1":  # keep only the light on its vertex
        from opencl_pathtracer_amd import structs as S
        t = np.zeros(len(sc.triangulation) - 5, S.Triangle)
        t[:] = sc.triangulation[:-5]
        sc.triangulation = t
    bvh_create(sc)
    print("light", sc.lights["position"][0].tolist(), "type", int(sc.lights["type"][0]))
    for it in range(iters):
        r = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it, strict=strict)
        o = O.oracle_render(sc, w, h, d, 1, first_iteration=it, default_arithmetic=not strict)
        for k, what in ((1, "box"), (2, "triangle")):
            diff = o[2][k].astype(np.int64) - r[2][k].astype(np.int64)
            print(f"iteration {it} {what} tests: oracle-only bins {np.nonzero(diff > 0)[0].tolist()}  reference-only bins {np.nonzero(diff < 0)[0].tolist()}")
        only = np.nonzero(o[2][1].astype(np.int64) - r[2][1].astype(np.int64) > 0)[0].tolist()
        only_t = np.nonzero(o[2][2].astype(np.int64) - r[2][2].astype(np.int64) > 0)[0].tolist()
        print("  image equal", np.array_equal(o[0].view(np.uint32), r[0].view(np.uint32)), "pixels that differ",
              np.argwhere((o[0].view(np.uint32) != r[0].view(np.uint32)).any(-1)).tolist()[:20])
        if not only:
            continue
        for y in range(h):
            for x in range(w):
                b, _ = O.oracle_trace(sc, w, h, d, x, y, it)
                if b and b[-1].n_bbx in only and b[-1].n_tri in only_t:
                    print(f"  pixel ({x}, {y}): {len(b)} bounces")
                    for i, q in enumerate(b):
                        tr = sc.triangulation[q.triangle_id]
                        print(f"      hit triangle S1 {tr['S1'].tolist()} S2 {tr['S2'].tolist()} S3 {tr['S3'].tolist()} N {tr['N'].tolist()}")
                        print(f"    bounce {i}: triangle {q.triangle_id} material {q.material_id} s {q.s:.6g} t {q.t:.6g} point {list(q.point)} "
                              f"out {list(q.out_dir)} box tests so far {q.n_bbx} triangle tests {q.n_tri}")


if __name__ == "__main__":
    main()
