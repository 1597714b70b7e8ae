"""Diagnostic (GPU box): traversal-work totals (sum of per-path counters) of reference default / strict / HIP
on variants of the matmix scene that keep the compiled specialisation (96x96, depth 8, 3 lights)."""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import oracle_ffi as O  # noqa
from opencl_pathtracer_amd import scenes, bvh_create, render_scene, structs as S  # noqa

case, w, h, d = "matmix_96x96_d8", 96, 96, 8
base = bvh_create(scenes.build("matmix", w, h))
k = np.arange(5000)


def totals(hist):
    return int((hist.astype(np.int64) * k).sum())


def variant(name, edit):
    sc = copy.copy(base)
    sc.materiaux = base.materiaux.copy()
    sc.lights = base.lights.copy()
    sc.triangulation = base.triangulation.copy()
    edit(sc)
    out = {}
    for label, fn in (("ref", lambda: O.ref_gpu_render(case, sc, w, h, d, 4)),
                      ("strict", lambda: O.ref_gpu_render(case, sc, w, h, d, 4, strict=True)),
                      ("hip", lambda: render_scene(sc, w, h, d, 4)[:3])):
        res = fn()
        dep, bbx, tri = res[2]
        out[label] = (totals(bbx), totals(tri), dep.tolist())
    print(f"{name:28s} bbx ref/strict/hip {out['ref'][0]} {out['strict'][0]} {out['hip'][0]} | tri "
          f"{out['ref'][1]} {out['strict'][1]} {out['hip'][1]} | depth-hist equal {out['ref'][2] == out['hip'][2]}")


def all_standard(sc):
    sc.materiaux["type"] = S.MAT_STANDART


def only(mt):
    def f(sc):
        t = sc.materiaux["type"]
        sc.materiaux["type"] = np.where(t == mt, mt, S.MAT_STANDART)
    return f


def lights_far(sc):
    sc.lights["power"] = 0  # same rays, no contribution


def flat_normals(sc):
    n = sc.triangulation["N"].copy()
    n[:, 3] = 0
    for f in ("N1", "N2", "N3"):
        sc.triangulation[f] = n


variant("original", lambda sc: None)
variant("all MAT_STANDART", all_standard)
variant("only GLASS kept", only(S.MAT_GLASS))
variant("only WATER kept", only(S.MAT_WATER))
variant("only VARNISHED kept", only(S.MAT_VARNHISHED))
variant("only METAL kept", only(S.MAT_METAL))
variant("flat vertex normals", flat_normals)
variant("flat normals + all STANDART", lambda sc: (flat_normals(sc), all_standard(sc)))
