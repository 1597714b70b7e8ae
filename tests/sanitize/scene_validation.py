"""Structural corruptions of valid scenes through ptmi_validate_scene (host-only): every index the device would follow.
Imported by tests/test_scene_validation.py (in-process, the product library) and run as a script with the library given on the
command line - the sanitizer build of csrc/scene_layout.cpp + csrc/bvh_build.cpp, libasan / libubsan preloaded.
A corruption must end in an error code or in an accepted scene; never in a fault, never in a read outside the arrays."""
import ctypes as C
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.simplefilter("ignore")
from opencl_pathtracer_amd import scenes, structs as S, backend  # noqa: E402

CORRUPTIONS = ["son_out_of_range", "son_cycle", "son_shared", "leaf_range", "leaf_count_huge", "cut_axis", "material_index", "texture_id",
               "texture_extent", "sky_texture", "uv_huge", "uv_nan", "lights_mismatch", "root_leaf_flag", "is_leaf_garbage", "texture_zero_size",
               "triangle_count_lie", "bvh_size_lie", "depth_chain"]


def corrupt(sc, kind, rs):
    """In place; returns overrides for the descriptor's sizes, or None when the scene has nothing of that kind to corrupt."""
    b, t, over = sc.bvh, sc.triangulation, {}
    inner = np.flatnonzero(b["isLeaf"] == 0)
    leaves = np.flatnonzero(b["isLeaf"] != 0)
    pick = lambda a: int(a[rs.randint(0, len(a))])
    if kind == "son_out_of_range" and len(inner):
        b[rs.choice(["son1Id", "son2Id"])][pick(inner)] = int(rs.choice([len(b), len(b) + 7, 2 ** 31, 2 ** 32 - 1]))
    elif kind == "son_cycle" and len(inner):
        i = pick(inner)
        b["son2Id"][i] = int(rs.choice([i, 0]))
    elif kind == "son_shared" and len(inner) > 1:
        i, j = pick(inner), pick(inner)
        b["son1Id"][i] = b["son1Id"][j]
    elif kind == "leaf_range":
        b["triangleStartIndex"][pick(leaves)] = int(rs.choice([len(t), len(t) - 1, 2 ** 31, 2 ** 32 - 2]))
    elif kind == "leaf_count_huge":
        b["nbTriangles"][pick(leaves)] = int(rs.choice([len(t) + 1, 2 ** 31, 2 ** 32 - 1]))
    elif kind == "cut_axis" and len(inner):
        b["cutAxis"][pick(inner)] = int(rs.choice([3, 255, 2 ** 31]))
    elif kind == "material_index":
        t[rs.choice(["materialWithPositiveNormalIndex", "materialWithNegativeNormalIndex"])][rs.randint(0, len(t))] = int(rs.choice([len(sc.materiaux), 2 ** 32 - 1]))
    elif kind == "texture_id":
        i = int(rs.randint(0, len(sc.materiaux)))
        sc.materiaux["isSimpleColor"][i] = 0
        sc.materiaux["textureId"][i] = int(rs.choice([len(sc.textures), -1, 2 ** 31 - 1]))
    elif kind == "texture_extent" and len(sc.textures):
        i = int(rs.randint(0, len(sc.textures)))
        sc.textures[rs.choice(["width", "height", "offset"])][i] = int(rs.choice([2 ** 16, 2 ** 31, 2 ** 32 - 1]))
    elif kind == "texture_zero_size" and len(sc.textures):
        sc.textures[rs.choice(["width", "height"])][int(rs.randint(0, len(sc.textures)))] = 0
    elif kind == "sky_texture":
        sc.sky["skyTextures"][int(rs.randint(0, 6))] = (int(rs.choice([0, 2 ** 16])), 3, int(rs.choice([0, 2 ** 32 - 1])))
    elif kind in ("uv_huge", "uv_nan"):
        textured = np.flatnonzero(sc.materiaux["isSimpleColor"][t["materialWithPositiveNormalIndex"]] == 0)
        if not len(textured):
            return None
        i = pick(textured)
        t[rs.choice(["UVP1", "UVN3"])][i] = (1e30, 0.5) if kind == "uv_huge" else (np.nan, np.inf)
    elif kind == "lights_mismatch":
        over["lights_size"] = len(sc.lights) + 1
    elif kind == "root_leaf_flag":
        b["isLeaf"][0] = 1
        b["nbTriangles"][0] = int(rs.choice([0, len(t), len(t) + 5]))
    elif kind == "is_leaf_garbage":
        b["isLeaf"][pick(np.arange(len(b)))] = int(rs.choice([2, 127, -1]))
    elif kind == "triangle_count_lie":
        over["triangulation_size"] = int(rs.choice([0, 1, max(1, len(t) // 2)]))
    elif kind == "bvh_size_lie":
        over["bvh_size"] = int(rs.choice([0, 1, max(1, len(b) // 2)]))
    elif kind == "depth_chain" and len(inner) > 2:
        for i in inner[:-1]:  # a chain: every inner node's second child is the next inner node (deeper than 30 for big trees, and shared)
            b["son2Id"][i] = i + 1
    elif kind not in ("lights_mismatch", "root_leaf_flag", "is_leaf_garbage", "triangle_count_lie", "bvh_size_lie", "leaf_range", "leaf_count_huge",
                      "material_index", "texture_id", "sky_texture"):
        return None  # (no inner node / no texture in this scene)
    return over


def run(lib, seeds=range(0, 24)):
    lib.ptmi_validate_scene.argtypes = [C.POINTER(backend.Config), C.POINTER(backend.SceneDesc)]
    lib.ptmi_bvh_create.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    outcomes = {}

    def validate(sc, over=None, sampler=S.JITTERED, ss=0):
        cfg = backend.Config(C.sizeof(backend.Config), 0, 64, 64, 8, sc.lightsSize, sampler, ss, 0)
        d, keep = backend.scene_desc(sc)
        for k, v in (over or {}).items():
            if k == "lights_size":
                cfg.lights_size = v - 1  # (the scene says one more than the configuration)
            setattr(d, k, v)
        rc = lib.ptmi_validate_scene(C.byref(cfg), C.byref(d))
        del keep
        return rc

    def with_tree(name):
        sc = scenes.build(name, 64, 64)
        t = np.ascontiguousarray(sc.triangulation)
        bvh = np.zeros(2 * len(t) - 1, S.Node)
        size, depth = C.c_uint32(0), C.c_uint32(0)
        assert lib.ptmi_bvh_create(t.ctypes.data, len(t), bvh.ctypes.data, C.byref(size), C.byref(depth)) == 0
        sc.triangulation, sc.bvh, sc.bvhMaxDepth = t, bvh[:size.value].copy(), depth.value
        return sc

    for seed in seeds:
        for name in (f"fuzz{seed}_l1", f"fuzz{seed}hr_l3", "matmix" if seed % 4 == 0 else "cornell"):
            assert validate(with_tree(name)) == 0, name
            assert validate(with_tree(name), sampler=S.RANDOM) == 0, name
            # boxes that do not bound, inverted, NaN / infinite faces, foreign split axes: structurally valid, so accepted
            assert validate(scenes.corrupt_tree(with_tree(name), seed)) == 0, name + " (corrupted tree)"
            rs = np.random.RandomState(31 * seed + len(name))
            for kind in CORRUPTIONS:
                sc = with_tree(name)
                over = corrupt(sc, kind, rs)
                if over is None:
                    continue
                rc = validate(sc, over)
                outcomes.setdefault(kind, set()).add(rc)
    return outcomes


if __name__ == "__main__":
    out = run(C.CDLL(sys.argv[1]))
    print("clean", {k: sorted(v) for k, v in out.items()})
