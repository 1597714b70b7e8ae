#!/usr/bin/env python3
"""Which element of a hostile fuzz scene makes the integrator and the reference kernel part?  (GPU box; needs oracle/_ref.)

usage: tools/diag_fuzz.py SEED [LIGHTS]
Builds scenes.fuzz_scene(SEED, hostile=True), then variants with one of its five hostile triangles (the last five: point,
pair, collinear, far, tiny) removed, or the light moved off its vertex, and prints the depth histograms of the reference kernel (default
build), the integrator and the CPU oracle for each."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, render_scene, structs as S  # noqa: E402


def main():
    seed = int(sys.argv[1])
    nl = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    case, w, h, d = ("feat_64x64_d8", 64, 64, 8) if nl == 1 else ("matmix_96x96_d8", 96, 96, 8)
    warnings.simplefilter("ignore")
    base = scenes.fuzz_scene(seed, w, h, n_lights=nl, hostile=True)
    n = len(base.triangulation)
    names = {n - 5: "point", n - 4: "pair", n - 3: "collinear", n - 2: "far", n - 1: "tiny"}
    variants = [("all", None, False)] + [("without " + names[i], i, False) for i in names] + [("light off the vertex", None, True)] + \
               [("only " + names[i], [j for j in names if j != i], False) for i in names] + [("only the light on a vertex", list(names), False)]
    for label, drop, move in variants:
        sc = scenes.fuzz_scene(seed, w, h, n_lights=nl, hostile=True)
        if drop is not None:
            keep = np.ones(n, bool)
            keep[drop] = False
            t = np.zeros(int(keep.sum()), S.Triangle)
            t[:] = sc.triangulation[keep]
            t["id"] = np.arange(len(t), dtype=np.uint32)
            sc.triangulation = t
        if move:
            sc.lights["position"][0][:3] += np.float32(0.37)
        if label.startswith("only") and "light" not in label:
            sc.lights["position"][0][:3] += np.float32(0.37)
        bvh_create(sc)
        strict = os.environ.get("STRICT") == "1"
        da = 0 if strict else 16
        r = O.ref_gpu_render(case, sc, w, h, d, 4, strict=strict)
        g = render_scene(sc, w, h, d, 4, flags=da)
        o = O.oracle_render(sc, w, h, d, 4, default_arithmetic=not strict)
        m = render_scene(sc, w, h, d, 4, flags=da | 2)  # the one-path-per-lane kernel
        tot = lambda x: [int((np.arange(len(x[2][k])) * x[2][k].astype(np.int64)).sum()) for k in (1, 2)]
        print(f"    box / triangle tests: ref {tot(r)} hip {tot(g)} oracle {tot(o)} one-path-per-lane {tot(m)}")
        same = np.array_equal(r[0].view(np.uint32), g[0].view(np.uint32))
        print(f"{label:24s} ref {r[2][0][:9].tolist()}  hip {g[2][0][:9].tolist()}  oracle {o[2][0][:9].tolist()}  one-path-per-lane {m[2][0][:9].tolist()} image equal ref {np.array_equal(r[0].view(np.uint32), m[0].view(np.uint32))}  wavefront image equal {same}"
              f"  nan ref/hip {int(np.isnan(r[0]).sum())}/{int(np.isnan(g[0]).sum())}", flush=True)


if __name__ == "__main__":
    main()
