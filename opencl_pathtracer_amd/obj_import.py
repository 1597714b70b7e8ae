"""Wavefront OBJ -> reference scene arrays, standing in for the Maya importer (which needs the Maya SDK).

What ``Maya/PathTracer_MayaImporter.cpp`` does to a mesh is kept: polygons are fanned into triangles
(``ImportMesh`` :177-308 triangulates through Maya), every triangle goes through ``Triangle_Create``
(``scenes.triangle_create``: AABB, N from the file's winding with w=1, lexicographic vertex order, vertex normals
normalised or replaced by N), one material per ``usemtl`` group mapped like the importer maps shaders
(:852-932: everything diffuse -> MAT_STANDART; a material whose name contains "phong"/"varnish" -> MAT_VARNHISHED,
"glass" -> MAT_GLASS, "water" -> MAT_WATER, "metal" -> MAT_METAL), colour from ``Kd`` of an optional .mtl
(black colours become 0.8 grey like :868-870).  Textures are not read (``isSimpleColor``); uv coordinates are kept so
a caller can attach textures.  The camera / lights / sky are the caller's (OBJ has none).
"""
import os

import numpy as np

from . import structs as S
from . import scenes
from .scenes import material_create, triangle_create, _concat_tris, _records


def _material_type(name):
    n = name.lower()
    for key, t in (("glass", S.MAT_GLASS), ("water", S.MAT_WATER), ("metal", S.MAT_METAL), ("phong", S.MAT_VARNHISHED),
                   ("varnish", S.MAT_VARNHISHED)):
        if key in n:
            return t
    return S.MAT_STANDART


def _read_mtl(path):
    colors, cur = {}, None
    if not os.path.exists(path):
        return colors
    for line in open(path, errors="ignore"):
        p = line.split()
        if not p:
            continue
        if p[0] == "newmtl":
            cur = p[1]
        elif p[0] == "Kd" and cur is not None:
            colors[cur] = tuple(float(x) for x in p[1:4])
    return colors


def load_obj(path, axis_permutation=(0, 1, 2)):
    """Returns (triangulation, materiaux, material_names).  axis_permutation=(2, 0, 1) reproduces the Maya
    importer's xyz -> zxy convention (MayaImporter.h:29-33) for y-up assets."""
    v, vt, vn = [], [], []
    faces = {}  # material -> list of (vi, ti, ni) triples per triangle
    mtl_colors, cur = {}, "default"
    for line in open(path, errors="ignore"):
        p = line.split()
        if not p or p[0].startswith("#"):
            continue
        if p[0] == "v":
            v.append([float(x) for x in p[1:4]])
        elif p[0] == "vt":
            vt.append([float(x) for x in p[1:3]])
        elif p[0] == "vn":
            vn.append([float(x) for x in p[1:4]])
        elif p[0] == "mtllib":
            mtl_colors.update(_read_mtl(os.path.join(os.path.dirname(path), p[1])))
        elif p[0] == "usemtl":
            cur = p[1]
        elif p[0] == "f":
            idx = []
            for tok in p[1:]:
                parts = (tok.split("/") + ["", ""])[:3]
                vi = int(parts[0])
                ti = int(parts[1]) if parts[1] else 0
                ni = int(parts[2]) if parts[2] else 0
                fix = lambda i, n: i - 1 if i > 0 else (n + i if i < 0 else -1)
                idx.append((fix(vi, len(v)), fix(ti, len(vt)), fix(ni, len(vn))))
            for k in range(1, len(idx) - 1):  # fan triangulation
                faces.setdefault(cur, []).append((idx[0], idx[k], idx[k + 1]))
    V = np.asarray(v, np.float32).reshape(-1, 3)[:, list(axis_permutation)]
    VT = np.asarray(vt, np.float32).reshape(-1, 2)
    VN = np.asarray(vn, np.float32).reshape(-1, 3)
    if len(VN):
        VN = VN[:, list(axis_permutation)]
    parts, mats, names = [], [], []
    for name, tris in faces.items():
        t = np.asarray(tris, np.int64)  # (n, 3 corners, 3 indices)
        pos = V[t[:, :, 0]]
        # degenerate triangles are dropped like Triangle_isValid (MayaImporter.cpp:934-941)
        ok = (np.any(pos[:, 0] != pos[:, 1], axis=1) & np.any(pos[:, 0] != pos[:, 2], axis=1) & np.any(pos[:, 1] != pos[:, 2], axis=1))
        t, pos = t[ok], pos[ok]
        if not len(t):
            continue
        uv = np.where((t[:, :, 1] >= 0)[..., None], VT[np.maximum(t[:, :, 1], 0)] if len(VT) else 0, 0).astype(np.float32) \
            if len(VT) else np.zeros((len(t), 3, 2), np.float32)
        if len(VN):
            nrm = np.where((t[:, :, 2] >= 0)[..., None], VN[np.maximum(t[:, :, 2], 0)], 0).astype(np.float32)
        else:
            nrm = None
        m = len(mats)
        parts.append(triangle_create(pos[:, 0], pos[:, 1], pos[:, 2], normals=nrm, uvp=uv, uvn=uv, mat_pos=m, mat_neg=m))
        kd = mtl_colors.get(name, (0.5, 0.5, 0.5))
        if all(c < 0.01 for c in kd):
            kd = (0.8, 0.8, 0.8)
        mats.append(material_create(_material_type(name), color=(kd[0], kd[1], kd[2], 0.0)))
        names.append(name)
    if not parts:
        raise ValueError(f"{path}: no triangles")  # the Maya importer throws on an empty scene too (:39)
    return _concat_tris(parts), _records(mats, S.Material), names


def scene_from_obj(path, width, height, axis_permutation=(0, 1, 2), fov_span=0.8):
    """A renderable Scene around an OBJ mesh: the camera looks at the mesh's bounding box from the -y side (z up, as in
    the built-in scenes), one point light above and behind the camera, a light-grey sky.  A convenience for
    examples/render.py - the reference gets all of this from its Maya scene."""
    tris, mats, _ = load_obj(path, axis_permutation)
    pts = np.concatenate([tris["S1"][:, :3], tris["S2"][:, :3], tris["S3"][:, :3]])
    lo, hi = pts.min(0), pts.max(0)
    centre, size = (lo + hi) / 2, float(np.linalg.norm(hi - lo))
    eye = centre + np.array([0.0, -1.6 * size, 0.35 * size])
    view = centre - eye
    view = view / np.linalg.norm(view)
    right = np.cross(view, [0.0, 0.0, 1.0])
    right = right / np.linalg.norm(right)
    up = np.cross(right, view)
    pos, d, r, u = scenes.camera(eye, view, right * fov_span, up * (fov_span * height / width))
    lights = scenes._records([scenes.light_point(eye + np.array([0.4 * size, 0.0, 1.2 * size]), power=6.0 * size * size)],
                             S.Light)
    sky, texels = scenes.no_sky((190, 200, 215, 255))
    return scenes.Scene(tris, lights, mats, np.zeros(0, S.Texture), texels, sky, pos, d, r, u, name=os.path.basename(path))
