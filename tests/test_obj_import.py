"""OBJ import goes through the same Triangle_Create conventions as the reference's Maya importer."""
import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import obj_import, scenes, bvh_create, structs as S

CUBE = """mtllib cube.mtl
v -1 -1 -1
v 1 -1 -1
v 1 1 -1
v -1 1 -1
v -1 -1 1
v 1 -1 1
v 1 1 1
v -1 1 1
vn 0 0 -1
vn 0 0 1
usemtl red_phong
f 1//1 4//1 3//1 2//1
usemtl floor
f 5//2 6//2 7//2 8//2
f 1 2 6 5
f 2 3 7 6
usemtl glass_pane
f 3 4 8 7
f 4 1 5 8
f 1 1 2
"""
MTL = "newmtl red_phong\nKd 0.8 0.1 0.1\nnewmtl floor\nKd 0 0 0\n"


def test_obj_cube(tmp_path, built):
    (tmp_path / "cube.obj").write_text(CUBE)
    (tmp_path / "cube.mtl").write_text(MTL)
    tris, mats, names = obj_import.load_obj(str(tmp_path / "cube.obj"))
    assert len(tris) == 12 and names == ["red_phong", "floor", "glass_pane"]  # quads fanned, degenerate face dropped
    assert list(mats["type"]) == [S.MAT_VARNHISHED, S.MAT_STANDART, S.MAT_GLASS]
    assert np.allclose(mats["simpleColor"][0], (0.8, 0.1, 0.1, 0)) and np.allclose(mats["simpleColor"][1][:3], 0.8)
    # Triangle_Create conventions
    assert (tris["S1"][:, 3] == 1).all() and (tris["N"][:, 3] == 1).all()
    for a, b in (("S1", "S2"), ("S2", "S3")):
        lex = [tuple(x[:3]) <= tuple(y[:3]) for x, y in zip(tris[a], tris[b])]
        assert all(lex)
    assert np.allclose(np.abs(tris["N"][:2, :3]), [[0, 0, 1], [0, 0, 1]])
    assert (tris["N1"][:2, 3] == 0).all()              # file normals are directions: w = 0
    assert (tris["N1"][4:, 3] == 1).all()              # no normals in the file: fallback to N (w = 1)
    # renderable end to end
    sc = scenes.cornell_box(32, 24)
    sc.triangulation, sc.materiaux = tris, mats
    sc.cameraPosition = np.array([0.3, -6, 0.2, 1], np.float32)
    sc.lights["position"][0] = (2, -4, 3, 1)
    sc.lights["power"][0] = 30
    bvh_create(sc)
    color, count, (dep, _, _), tot = O.oracle_render(sc, 32, 24, 4, 2)
    assert np.isfinite(color).all() and dep[1:].sum() > 0 and tot["paths"] == 32 * 24 * 2
    zxy, _, _ = obj_import.load_obj(str(tmp_path / "cube.obj"), axis_permutation=(2, 0, 1))
    assert np.allclose(np.abs(zxy["N"][0, :3]), [1, 0, 0])  # Maya's xyz -> zxy


def test_empty_obj_is_an_error(tmp_path):
    (tmp_path / "e.obj").write_text("v 0 0 0\n")
    with pytest.raises(ValueError):
        obj_import.load_obj(str(tmp_path / "e.obj"))


def test_scene_from_obj_frames_the_mesh(tmp_path, built):
    """The convenience wrapper used by examples/render.py: camera outside the mesh looking at it, oracle renders hits."""
    (tmp_path / "cube.obj").write_text(CUBE)
    (tmp_path / "cube.mtl").write_text(MTL)
    sc = bvh_create(obj_import.scene_from_obj(str(tmp_path / "cube.obj"), 32, 24))
    color, count, (dep, _, _), totals = O.oracle_render(sc, 32, 24, 3, 2)
    assert totals["surface_hits"] > 0 and np.isfinite(color).all() and (count == 2).all()
    centre = color[8:16, 12:20, :3].mean()
    assert centre > 0  # the mesh is in the middle of the frame and lit
