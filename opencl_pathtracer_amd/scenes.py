"""Scene construction for the integrator: the conventions the reference's importers follow.

The reference ships no scene assets and its only live importer needs the Maya SDK
(``Maya/PathTracer_MayaImporter.cpp``).  What that importer *does* to raw geometry is
the de-facto spec of valid scene data, so it is restated here with numpy (vectorised,
float32, one rounding per operation like the host code):

* ``triangle_create``  <- ``Triangle_Create`` (MayaImporter.cpp:943-1047): AABB, geometric normal
  from the ORIGINAL winding with ``w = 1`` (host ``cross`` returns w=1, Utils.h:131), vertices
  sorted lexicographically together with their attributes, vertex normals normalised or replaced
  by ``N`` when degenerate.
* points carry ``w = 1``, directions ``w = 0`` (MayaImporter.h:29-37).
* lights / materials / "no sky" / cube-map cross  <- MayaImporter.cpp:819-850, 852-932, 695-817.

The generators below build the BASELINE.json workloads (Cornell box, N random triangles)
and a material/texture/light mix used by the parity tests.  Every scene is a ``Scene``:
the subset of the reference's ``GlobalVars`` (PathTracer_Structs.h:145-189) the device
backend reads.
"""
import re
from dataclasses import dataclass, field

import numpy as np

from . import structs as S

f32 = np.float32


@dataclass
class Scene:
    triangulation: np.ndarray
    lights: np.ndarray
    materiaux: np.ndarray
    textures: np.ndarray
    texturesData: np.ndarray
    sky: np.ndarray
    cameraPosition: np.ndarray
    cameraDirection: np.ndarray
    cameraRight: np.ndarray
    cameraUp: np.ndarray
    bvh: np.ndarray = None
    bvhMaxDepth: int = 0
    name: str = ""
    meta: dict = field(default_factory=dict)

    @property
    def lightsSize(self):
        return len(self.lights)


# --------------------------------------------------------------------------- helpers

def _f4(xyz, w):
    xyz = np.asarray(xyz, dtype=f32)
    out = np.empty(xyz.shape[:-1] + (4,), dtype=f32)
    out[..., :3] = xyz
    out[..., 3] = w
    return out


def _dot3(a, b):
    # host dot(), Utils.h:130: (x*x) + (y*y) + (z*z), float
    return (a[..., 0] * b[..., 0] + a[..., 1] * b[..., 1]) + a[..., 2] * b[..., 2]


def _cross3(a, b):
    # host cross(), Utils.h:131 (w handled by the caller)
    return np.stack([a[..., 1] * b[..., 2] - a[..., 2] * b[..., 1],
                     a[..., 2] * b[..., 0] - a[..., 0] * b[..., 2],
                     a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]], axis=-1).astype(f32)


def triangle_create(s1, s2, s3, normals=None, uvp=None, uvn=None, mat_pos=0, mat_neg=None, first_id=0):
    """Vectorised ``Triangle_Create`` (MayaImporter.cpp:943-1047).

    s1,s2,s3: (n,3) vertex positions in the caller's winding.  normals: (n,3,3) per-vertex
    normals (zero rows trigger the reference's fallback to N, which carries w=1) or None for
    flat shading with proper w=0 normals.  uvp/uvn: (n,3,2).  Returns a Triangle array.
    """
    s = np.stack([np.asarray(s1, f32), np.asarray(s2, f32), np.asarray(s3, f32)], axis=1)  # (n,3,3)
    n = s.shape[0]
    tri = np.zeros(n, dtype=S.Triangle)

    p = _f4(s, 1.0)  # points: w = 1
    # BoundingBox_Create, Structs.h:197-205 (all four components)
    pmin = np.minimum(np.minimum(p[:, 0], p[:, 1]), p[:, 2])
    pmax = np.maximum(np.maximum(p[:, 0], p[:, 1]), p[:, 2])
    tri["AABB"]["pMin"] = pmin
    tri["AABB"]["pMax"] = pmax
    tri["AABB"]["centroid"] = (pmin + pmax) / f32(2)
    tri["AABB"]["isEmpty"] = 0

    # N = normalize(cross(s2-s1, s3-s1)) with the ORIGINAL order; host normalize keeps w (=1)
    c = _cross3(s[:, 1] - s[:, 0], s[:, 2] - s[:, 0])
    length = np.sqrt(_dot3(c, c)).astype(f32)
    N = _f4(c / length[:, None], 1.0)
    tri["N"] = N

    if normals is None:
        nrm = np.repeat(N[:, None, :3], 3, axis=1)
    else:
        nrm = np.asarray(normals, f32)
    nrm4 = _f4(nrm, 0.0)  # directions: w = 0
    uvp = np.zeros((n, 3, 2), f32) if uvp is None else np.asarray(uvp, f32)
    uvn = uvp if uvn is None else np.asarray(uvn, f32)

    # lexicographic vertex order (Vector_LexLessThan cascade, :964-1017) == a sort on (x,y,z)
    order = np.lexsort((s[:, :, 2], s[:, :, 1], s[:, :, 0]), axis=1)
    rows = np.arange(n)[:, None]
    p = p[rows, order]
    nrm4 = nrm4[rows, order]
    uvp = uvp[rows, order]
    uvn = uvn[rows, order]

    # vertex normals: length < 0.5 -> N (w=1!), else divide all four components (:1028-1033)
    ln = np.sqrt(_dot3(nrm4, nrm4)).astype(f32)
    bad = ln < f32(0.5)
    with np.errstate(divide="ignore", invalid="ignore"):
        nn = nrm4 / ln[..., None]
    nn = np.where(bad[..., None], N[:, None, :], nn).astype(f32)

    for k, name in enumerate(("S1", "S2", "S3")):
        tri[name] = p[:, k]
    for k, name in enumerate(("N1", "N2", "N3")):
        tri[name] = nn[:, k]
    for k, name in enumerate(("UVP1", "UVP2", "UVP3")):
        tri[name] = uvp[:, k]
    for k, name in enumerate(("UVN1", "UVN2", "UVN3")):
        tri[name] = uvn[:, k]
    tri["materialWithPositiveNormalIndex"] = mat_pos
    tri["materialWithNegativeNormalIndex"] = mat_pos if mat_neg is None else mat_neg
    tri["id"] = np.arange(first_id, first_id + n, dtype=np.uint32)
    return tri


def material_create(mtype=S.MAT_STANDART, color=(0.5, 0.5, 0.5, 0.0), texture_id=-1, opacity=1.0):
    """Material_Create overloads (MayaImporter.cpp:852-932): textured iff textureId >= 0."""
    m = np.zeros((), dtype=S.Material)
    m["type"] = mtype
    m["simpleColor"] = np.asarray(color, f32)
    m["textureId"] = texture_id if texture_id >= 0 else 0
    m["isSimpleColor"] = 1 if texture_id < 0 else 0
    m["hasAlphaMap"] = 0
    m["opacity"] = opacity
    return m


def light_point(position, color=(1, 1, 1, 1), power=1.0):
    l = np.zeros((), dtype=S.Light)  # Light_Create(MFnPointLight), MayaImporter.cpp:819-828
    l["type"] = S.LIGHT_POINT
    l["color"] = np.asarray(color, f32)
    l["direction"] = (0, 0, 0, 1)
    l["position"] = _f4(position, 1.0)
    l["power"] = power
    return l


def light_directional(direction, color=(1, 1, 1, 1), power=1.0):
    l = np.zeros((), dtype=S.Light)  # MayaImporter.cpp:830-839
    d = np.asarray(direction, f32)
    d = d / np.sqrt(_dot3(d, d)).astype(f32)
    l["type"] = S.LIGHT_DIRECTIONNAL
    l["color"] = np.asarray(color, f32)
    l["direction"] = _f4(d, 0.0)
    l["position"] = (0, 0, 0, 1)
    l["power"] = power
    return l


def light_spot(position, direction, cone_angle, penumbra_angle, color=(1, 1, 1, 1), intensity=1.0):
    l = np.zeros((), dtype=S.Light)  # MayaImporter.cpp:841-850
    d = np.asarray(direction, f32)
    d = d / np.sqrt(_dot3(d, d)).astype(f32)
    l["type"] = S.LIGHT_SPOT
    l["color"] = np.asarray(color, f32)
    l["cosOfInnerFallOffAngle"] = f32(np.cos(cone_angle / 2))
    l["cosOfOuterFallOffAngle"] = f32(np.cos(cone_angle / 2 + penumbra_angle))
    l["direction"] = _f4(d, 0.0)
    l["position"] = _f4(position, 1.0)
    l["power"] = 3 * intensity
    return l


def no_sky(rgba=(0, 0, 0, 0)):
    """LoadSkyAndAllocateTextureMemory(loadSky=false), MayaImporter.cpp:695-719: six 1x1 faces at texel 0."""
    sky = np.zeros((), dtype=S.Sky)
    sky["cosRotationAngle"] = 1
    sky["sinRotationAngle"] = 0
    sky["groundScale"] = 1
    for i in range(6):
        sky["skyTextures"][i] = (1, 1, 0)
    texels = np.array([rgba], dtype=np.uint8)
    return sky, texels


def sky_from_cross(bgr, first_texel=0):
    """LoadSkyAndAllocateTextureMemory(loadSky=true), MayaImporter.cpp:720-817: unpack a horizontal-cross cube map
    (4 x 3 faces: the top face above and the ground face below the second of four side faces) into the six sky
    textures.  ``bgr`` is uint8[3*fh, 4*fw, 3] in the byte order of the importer's BMP buffer (B, G, R); texels come out
    as (r, g, b, 255), faces in the order top, four sides left to right, ground, each ``fw*fh`` texels, the first
    at ``first_texel`` of texturesData.  Returns (sky record, texels uint8[6*fw*fh, 4])."""
    bgr = np.asarray(bgr, np.uint8)
    full_h, full_w = bgr.shape[:2]
    fw, fh = full_w // 4, full_h // 3
    if fw == 0 or fh == 0:
        raise ValueError("cross image smaller than 4x3")
    blocks = [bgr[0:fh, fw:2 * fw]] + [bgr[fh:2 * fh, i * fw:(i + 1) * fw] for i in range(4)] + [bgr[2 * fh:3 * fh, fw:2 * fw]]
    texels = np.empty((6 * fw * fh, 4), np.uint8)
    sky = np.zeros((), dtype=S.Sky)
    sky["cosRotationAngle"] = 1
    sky["sinRotationAngle"] = 0
    sky["groundScale"] = 1
    for i, blk in enumerate(blocks):
        t = texels[i * fw * fh:(i + 1) * fw * fh]
        t[:, :3] = blk.reshape(-1, 3)[:, ::-1]  # b,g,r -> r,g,b
        t[:, 3] = 255
        sky["skyTextures"][i] = (fw, fh, first_texel + i * fw * fh)
    return sky, texels


def camera(position, direction, right, up):
    return _f4(position, 1.0), _f4(direction, 0.0), _f4(right, 0.0), _f4(up, 0.0)


def camera_from_film(eye, view, up, right, focal_length_mm, aperture_x_inch, aperture_y_inch, maya_axes=False):
    """SetCam, MayaImporter.cpp:59-101: unit view / up / right directions and film data -> the four camera vectors the
    kernel takes.  Right and Up are scaled by aperture / focal length (apertures come in inches, the focal length in
    millimetres) so that the sample offsets of +-0.5 span the film; with ``maya_axes`` the importer's xyz -> zxy
    permutation (MayaImporter.h:29-37) is applied as it is to everything coming out of Maya."""
    perm = (lambda v: np.asarray(v, np.float64)[[2, 0, 1]]) if maya_axes else (lambda v: np.asarray(v, np.float64))
    r = np.asarray(right, np.float64) * (aperture_x_inch * 25.4 / focal_length_mm)
    u = np.asarray(up, np.float64) * (aperture_y_inch * 25.4 / focal_length_mm)
    return camera(perm(eye), perm(view), perm(r), perm(u))


def _records(items, dtype):
    """Array of struct records with ZEROED padding bytes (np.array(list_of_records, dtype) leaves them undefined,
    and scene files / digests should not depend on heap garbage)."""
    out = np.zeros(len(items), dtype=dtype)
    for i, it in enumerate(items):
        for name in dtype.names:
            out[name][i] = it[name]
    return out


def _quad(a, b, c, d):
    """two triangles a-b-c, a-c-d"""
    a, b, c, d = (np.asarray(v, f32) for v in (a, b, c, d))
    return np.array([a, a]), np.array([b, c]), np.array([c, d])


def _concat_tris(parts):
    # not np.concatenate: numpy "promotes" padded struct dtypes to a packed layout
    t = np.zeros(sum(len(p) for p in parts), dtype=S.Triangle)
    k = 0
    for p in parts:
        assert p.dtype == S.Triangle
        t[k:k + len(p)] = p
        k += len(p)
    t["id"] = np.arange(len(t), dtype=np.uint32)
    return t


# --------------------------------------------------------------------------- BASELINE workloads

def cornell_box(width, height):
    """BASELINE configs 1-2: Cornell box, 32 triangles, one point light under a (non-emissive) lamp quad.

    555-unit scale, z up.  The reference has no emissive materials: light comes from Light[] and sky only.
    """
    WHITE, RED, GREEN = 0, 1, 2
    mats = _records([material_create(color=(0.73, 0.73, 0.73, 0)), material_create(color=(0.65, 0.05, 0.05, 0)),
                     material_create(color=(0.12, 0.45, 0.15, 0))], S.Material)
    parts = []

    def add_quad(a, b, c, d, mat):
        s1, s2, s3 = _quad(a, b, c, d)
        parts.append(triangle_create(s1, s2, s3, mat_pos=mat))

    L = 555.0
    add_quad((0, 0, 0), (L, 0, 0), (L, L, 0), (0, L, 0), WHITE)          # floor
    add_quad((0, 0, L), (0, L, L), (L, L, L), (L, 0, L), WHITE)          # ceiling
    add_quad((0, L, 0), (L, L, 0), (L, L, L), (0, L, L), WHITE)          # back wall
    add_quad((0, 0, 0), (0, L, 0), (0, L, L), (0, 0, L), RED)            # left wall
    add_quad((L, 0, 0), (L, 0, L), (L, L, L), (L, L, 0), GREEN)          # right wall

    def add_block(corners, h):
        c = [np.array([x, y, 0.0], f32) for x, y in corners]
        t = [v + np.array([0, 0, h], f32) for v in c]
        add_quad(t[0], t[1], t[2], t[3], WHITE)                           # top
        for i in range(4):
            j = (i + 1) % 4
            add_quad(c[i], c[j], t[j], t[i], WHITE)                       # sides

    add_block([(130, 65), (82, 225), (240, 272), (290, 114)], 165.0)      # short block
    add_block([(423, 247), (265, 296), (314, 456), (472, 406)], 330.0)    # tall block
    add_quad((213, 227, 554.9), (343, 227, 554.9), (343, 332, 554.9), (213, 332, 554.9), WHITE)  # lamp quad

    tris = _concat_tris(parts)
    assert len(tris) == 32
    lights = _records([light_point((278.0, 279.5, 420.0), power=120000.0)], S.Light)
    sky, texels = no_sky()
    span = 0.75
    pos, d, r, u = camera((278.0, -800.0, 273.0), (0, 1, 0), (span, 0, 0), (0, 0, span * height / width))
    return Scene(tris, lights, mats, np.zeros(0, S.Texture), texels, sky, pos, d, r, u, name="cornell")


def random_triangles(n, width, height, seed=12345):
    """BASELINE config 3: n random triangles, centres U[-5,5]^3, vertices centre + U[-0.1,0.1]^3.

    Generator pinned here: numpy MT19937 ``RandomState(seed)``; first ``uniform(-5,5,(n,3))`` for
    the centres, then ``uniform(-0.1,0.1,(n,3,3))`` for the vertex offsets (vertex-major), both
    rounded to float32 before the add.
    """
    rs = np.random.RandomState(seed)
    centres = rs.uniform(-5.0, 5.0, (n, 3)).astype(f32)
    offs = rs.uniform(-0.1, 0.1, (n, 3, 3)).astype(f32)
    v = (centres[:, None, :] + offs).astype(f32)
    tris = triangle_create(v[:, 0], v[:, 1], v[:, 2], mat_pos=0)
    mats = _records([material_create(color=(0.7, 0.7, 0.7, 0))], S.Material)
    lights = _records([light_point((-9.0, 3.0, 6.0), power=60.0)], S.Light)
    sky, texels = no_sky((200, 200, 200, 255))
    pos, d, r, u = camera((-14.0, 0.0, 0.0), (1, 0, 0), (0, 1, 0), (0, 0, height / width))
    return Scene(tris, lights, mats, np.zeros(0, S.Texture), texels, sky, pos, d, r, u, name=f"tris{n}",
                 meta={"seed": seed})


# --------------------------------------------------------------------------- parity-test mix

def _icosphere(subdiv):
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    f = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    v = [np.array(p, np.float64) / np.linalg.norm(p) for p in v]
    for _ in range(subdiv):
        cache, nf = {}, []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = v[a] + v[b]
                v.append(m / np.linalg.norm(m))
                cache[key] = len(v) - 1
            return cache[key]

        for a, b, c in f:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        f = nf
    return np.array(v), np.array(f)


def _sphere_tris(center, radius, subdiv, mat_pos, mat_neg=None, smooth=True):
    v, f = _icosphere(subdiv)
    p = (v * radius + np.asarray(center)).astype(f32)
    nrm = v.astype(f32)
    s = p[f]  # (n,3,3)
    normals = nrm[f] if smooth else None
    # spherical uv
    uv = np.stack([np.arctan2(v[:, 1], v[:, 0]) / (2 * np.pi) + 0.5, np.arccos(np.clip(v[:, 2], -1, 1)) / np.pi],
                  axis=-1).astype(f32)
    return triangle_create(s[:, 0], s[:, 1], s[:, 2], normals=normals, uvp=uv[f], uvn=(uv[f] * f32(3.0)).astype(f32),
                           mat_pos=mat_pos, mat_neg=mat_neg)


def _checker(w, h, c0, c1, cell=4):
    yy, xx = np.mgrid[0:h, 0:w]
    m = ((xx // cell + yy // cell) % 2).astype(bool)
    t = np.where(m[..., None], np.array(c1, np.uint8), np.array(c0, np.uint8)).astype(np.uint8)
    return t.reshape(-1, 4)


def _gradient(w, h, rgb0, rgb1):
    yy, xx = np.mgrid[0:h, 0:w]
    a = ((xx + yy) / float(w + h - 2))[..., None]
    t = np.concatenate([(1 - a) * np.array(rgb0) + a * np.array(rgb1), np.full((h, w, 1), 255.0)], axis=-1)
    return np.clip(t, 0, 255).astype(np.uint8).reshape(-1, 4)


def material_mix(width, height):
    """Parity-test scene touching every branch the kernel has: all five material types, textured and
    two-sided materials, smooth and fallback (w=1) vertex normals, point + spot + directional lights,
    a six-face cubemap sky.  Not a BASELINE workload."""
    texels = []
    textures = []

    def add_tex(img, w, h):
        off = sum(len(t) for t in texels)
        texels.append(img)
        textures.append((w, h, off))
        return len(textures) - 1

    sky = np.zeros((), dtype=S.Sky)
    sky["cosRotationAngle"] = f32(np.cos(0.3))
    sky["sinRotationAngle"] = f32(np.sin(0.3))
    sky["groundScale"] = 1
    face_cols = [((40, 40, 60), (90, 90, 140)), ((200, 120, 60), (250, 220, 160)), ((60, 140, 200), (160, 220, 250)),
                 ((120, 200, 120), (220, 250, 220)), ((200, 200, 80), (250, 250, 200)), ((120, 160, 250), (230, 240, 255))]
    for i, (c0, c1) in enumerate(face_cols):
        off = sum(len(t) for t in texels)
        texels.append(_gradient(16, 16, c0, c1))
        sky["skyTextures"][i] = (16, 16, off)

    t_check = add_tex(_checker(32, 32, (230, 230, 230, 0), (40, 40, 160, 0)), 32, 32)
    t_wood = add_tex(_gradient(24, 8, (150, 90, 40), (220, 170, 90)), 24, 8)

    M = dict(floor=0, red=1, glass=2, water=3, varnish_tex=4, metal=5, varnish=6, blue=7, white=8)
    mats = _records([
        material_create(S.MAT_STANDART, texture_id=t_check),
        material_create(S.MAT_STANDART, color=(0.8, 0.2, 0.15, 0)),
        material_create(S.MAT_GLASS, color=(0.9, 0.95, 1.0, 0), opacity=0.1),
        material_create(S.MAT_WATER, color=(0.5, 0.7, 0.9, 0)),
        material_create(S.MAT_VARNHISHED, texture_id=t_wood),
        material_create(S.MAT_METAL, color=(0.9, 0.9, 0.9, 0)),
        material_create(S.MAT_VARNHISHED, color=(0.2, 0.6, 0.3, 0)),
        material_create(S.MAT_STANDART, color=(0.2, 0.3, 0.8, 0)),
        material_create(S.MAT_STANDART, color=(0.8, 0.8, 0.8, 0)),
    ], S.Material)

    parts = []
    # textured floor, 8x8 quads with uv tiling
    g = 8
    xs = np.linspace(-6, 6, g + 1)
    for i in range(g):
        for j in range(g):
            a, b, c, d = (xs[i], xs[j], 0), (xs[i + 1], xs[j], 0), (xs[i + 1], xs[j + 1], 0), (xs[i], xs[j + 1], 0)
            s1, s2, s3 = _quad(a, b, c, d)
            uv = np.array([[(i, j), (i + 1, j), (i + 1, j + 1)], [(i, j), (i + 1, j + 1), (i, j + 1)]], f32) * f32(0.5)
            parts.append(triangle_create(s1, s2, s3, uvp=uv, uvn=uv, mat_pos=M["floor"], mat_neg=M["blue"]))
    # back wall: two-sided, different materials, fallback normals (zero -> N with w=1)
    s1, s2, s3 = _quad((-6, 6, 0), (6, 6, 0), (6, 6, 5), (-6, 6, 5))
    parts.append(triangle_create(s1, s2, s3, normals=np.zeros((2, 3, 3), f32), mat_pos=M["red"], mat_neg=M["white"]))
    # side wall
    s1, s2, s3 = _quad((-6, -6, 0), (-6, 6, 0), (-6, 6, 5), (-6, -6, 5))
    parts.append(triangle_create(s1, s2, s3, mat_pos=M["varnish"], mat_neg=M["varnish"]))
    parts.append(_sphere_tris((-2.5, 0.5, 1.3), 1.3, 2, M["glass"]))
    parts.append(_sphere_tris((1.0, 2.0, 1.0), 1.0, 2, M["varnish_tex"]))
    parts.append(_sphere_tris((2.8, -1.5, 0.8), 0.8, 1, M["metal"], smooth=False))
    parts.append(_sphere_tris((0.0, -2.5, 0.7), 0.7, 2, M["red"], mat_neg=M["blue"]))
    # water sheet over part of the floor
    s1, s2, s3 = _quad((-6, -6, 0.4), (0, -6, 0.4), (0, -1, 0.4), (-6, -1, 0.4))
    parts.append(triangle_create(s1, s2, s3, mat_pos=M["water"], mat_neg=M["water"]))
    tris = _concat_tris(parts)

    lights = _records([
        light_point((3.0, -4.0, 6.0), color=(1, 0.95, 0.9, 1), power=40.0),
        light_spot((-4.0, -4.0, 7.0), (0.5, 0.6, -1.0), cone_angle=0.7, penumbra_angle=0.3, color=(0.9, 0.9, 1, 1),
                   intensity=1.5),
        light_directional((-0.3, 0.4, -1.0), color=(1, 1, 0.9, 1), power=0.6),
    ], S.Light)
    span = 0.9
    pos, d, r, u = camera((0.5, -11.0, 3.5), (0, 1, -0.22), (span, 0, 0), (0, 0.22 * span * height / width, span * height / width))
    return Scene(tris, lights, mats, np.array(textures, dtype=S.Texture) if textures else np.zeros(0, S.Texture), np.concatenate(texels), sky, pos, d, r, u,
                 name="matmix")


# --------------------------------------------------------------------------- the stand-in for BASELINE configs[4]

def _grid_mesh(P, Nrm, UV, mat_pos, mat_neg=None, uv_night_scale=1.0):
    """Triangles of a (nu+1) x (nv+1) vertex grid (two per cell, the cell's shorter diagonal is not chosen: a mesh as a
    modeller's tessellation leaves it): per-vertex normals and uv as the importer copies them from the mesh."""
    nu, nv = P.shape[0] - 1, P.shape[1] - 1
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    i, j = i.ravel(), j.ravel()
    a, b, c, d = (i, j), (i + 1, j), (i + 1, j + 1), (i, j + 1)
    corner = lambda A, k: A[k[0], k[1]]
    tri_ix = [(a, b, c), (a, c, d)]
    s = np.concatenate([np.stack([corner(P, k) for k in t], axis=1) for t in tri_ix]).astype(f32)
    n = np.concatenate([np.stack([corner(Nrm, k) for k in t], axis=1) for t in tri_ix]).astype(f32)
    uv = np.concatenate([np.stack([corner(UV, k) for k in t], axis=1) for t in tri_ix]).astype(f32)
    # Triangle_isValid (MayaImporter.cpp:934-941): triangles with two equal vertices are not imported
    ok = (np.any(s[:, 0] != s[:, 1], axis=-1) & np.any(s[:, 0] != s[:, 2], axis=-1) & np.any(s[:, 1] != s[:, 2], axis=-1))
    s, n, uv = s[ok], n[ok], uv[ok]
    return triangle_create(s[:, 0], s[:, 1], s[:, 2], normals=n, uvp=uv, uvn=(uv * f32(uv_night_scale)).astype(f32),
                           mat_pos=mat_pos, mat_neg=mat_neg)


def _cube_sphere(center, radius, n, mat_pos, mat_neg=None, uv_tiles=1.0):
    """A sphere as six n x n patches of a cube pushed onto it (no poles, no zero-area triangles), smooth normals."""
    t = np.linspace(-1.0, 1.0, n + 1)
    u, v = np.meshgrid(t, t, indexing="ij")
    one = np.ones_like(u)
    faces = [(one, u, v), (-one, v, u), (v, one, u), (u, -one, v), (u, v, one), (v, u, -one)]
    parts = []
    for k, (x, y, z) in enumerate(faces):
        d = np.stack([x, y, z], axis=-1)
        d = d / np.linalg.norm(d, axis=-1, keepdims=True)
        uv = np.stack([(u + 1) * 0.5 * uv_tiles + 0.37 * k, (v + 1) * 0.5 * uv_tiles + 0.21 * k], axis=-1)
        parts.append(_grid_mesh(d * radius + np.asarray(center, np.float64), d, uv, mat_pos, mat_neg, uv_night_scale=2.0))
    return parts


def _noise_texture(rs, size, octaves, c0, c1, cell=0, alpha=0):
    """size x size RGBA texels: value noise (bilinear, `octaves` octaves) between two colours, optionally under a grid of
    `cell`-texel tiles with darker joints - what a photograph-based texture map is to the texel fetch: no two neighbours
    equal."""
    img = np.zeros((size, size))
    amp, total = 1.0, 0.0
    for o in range(octaves):
        g = 4 << o
        lat = rs.rand(g + 1, g + 1)
        x = np.linspace(0, g, size, endpoint=False)
        xi, xf = x.astype(int), x - x.astype(int)
        xf = xf * xf * (3 - 2 * xf)
        a = lat[xi][:, xi] * (1 - xf)[None, :] + lat[xi][:, xi + 1] * xf[None, :]
        b = lat[xi + 1][:, xi] * (1 - xf)[None, :] + lat[xi + 1][:, xi + 1] * xf[None, :]
        img += amp * (a * (1 - xf)[:, None] + b * xf[:, None])
        total += amp
        amp *= 0.55
    img /= total
    img = img + rs.uniform(-0.04, 0.04, img.shape)  # grain
    if cell:
        yy, xx = np.mgrid[0:size, 0:size]
        joint = ((xx % cell) < 3) | (((yy + (xx // cell % 2) * (cell // 2)) % (cell // 2)) < 3)
        img = np.where(joint, img * 0.45, img)
    img = np.clip(img, 0, 1)[..., None]
    rgb = (1 - img) * np.array(c0, np.float64) + img * np.array(c1, np.float64)
    out = np.empty((size * size, 4), np.uint8)
    out[:, :3] = np.clip(rgb, 0, 255).astype(np.uint8).reshape(-1, 3)
    out[:, 3] = alpha
    return out


MAYALIKE_FULL = dict(terrain=560, sphere=64, torus=(256, 128), pond=128, wall=(192, 64))   # 1,094,144 triangles
MAYALIKE_SMALL = dict(terrain=48, sphere=8, torus=(32, 16), pond=12, wall=(8, 4))           # 10,720 triangles: the CPU checker's size


def maya_like(width, height, detail=None, texture_size=1024, sky_face=512):
    """Stand-in for BASELINE configs[4] ("Maya-imported textured scene"; SURVEY 8d Config 5) - the importer needs the Maya SDK
    and the reference ships no scene, so this builds what PathTracerMayaImporter would hand over for a modelled outdoor set,
    following its conventions record by record:

    * tessellated OBJECTS with shared smooth vertex normals and uv sets, not soup (ImportMesh, MayaImporter.cpp:177-308):
      a 40 x 40 unit terrain of rolling hills (627 k triangles), six spheres, a torus, the rippled surface of a pond, three
      sculpted brick walls;
    * materials as CreateMaterials emits them (:606-690, :852-932): [0] the default grey for unknown shaders, Lambert ->
      MAT_STANDART and Phong -> MAT_VARNHISHED, with and without a file texture; plus one each of the types the importer never
      writes but the kernel implements (glass, water, metal), so the general shading code is all live;
    * FOUR file textures of 1024 x 1024 RGBA texels (16 MB) behind the six 512 x 512 faces of the cube-map sky, which
      comes first in texturesData as LoadSkyAndAllocateTextureMemory lays it out (:695-817), unpacked from a 2048 x 1536
      horizontal cross by sky_from_cross;
    * a point, a spot and a directional light (Light_Create, :819-850);
    * z up (the importer's xyz -> zxy permutation applied), a 35 mm film-back camera through camera_from_film (SetCam, :59-101).
    ``detail``: tessellation counts (MAYALIKE_FULL by default; MAYALIKE_SMALL keeps the same records at the CPU checker's size).
    """
    det = dict(MAYALIKE_FULL if detail is None else detail)
    rs = np.random.RandomState(20240416)
    # ---- sky: a horizontal cross, B G R bytes as the importer's BMP buffer holds them
    fw = fh = int(sky_face)
    cross = np.zeros((3 * fh, 4 * fw, 3), np.uint8)
    yy, xx = np.mgrid[0:3 * fh, 0:4 * fw]
    height_in_sky = 1.0 - yy / (3.0 * fh - 1)
    clouds = (np.sin(xx * (2 * np.pi / (4 * fw)) * 7 + 3 * np.sin(yy * 0.011)) * np.sin(yy * 0.017 + 1.3) * 0.5 + 0.5) ** 3
    clouds = clouds + rs.uniform(0, 0.05, clouds.shape)
    sky_rgb = np.stack([90 + 120 * clouds + 40 * (1 - height_in_sky), 140 + 90 * clouds + 20 * (1 - height_in_sky),
                        235 - 40 * height_in_sky + 20 * clouds], axis=-1)
    ground = yy >= 2 * fh
    sky_rgb = np.where(ground[..., None], np.stack([70 + 30 * clouds, 90 + 40 * clouds, 50 + 20 * clouds], axis=-1), sky_rgb)
    cross[...] = np.clip(sky_rgb, 0, 255).astype(np.uint8)[..., ::-1]
    sky, sky_texels = sky_from_cross(cross, first_texel=0)
    sky["cosRotationAngle"], sky["sinRotationAngle"] = f32(np.cos(0.4)), f32(np.sin(0.4))
    texels = [sky_texels]
    textures = []

    def add_tex(img):
        off = sum(len(t) for t in texels)
        texels.append(img)
        textures.append((texture_size, texture_size, off))
        return len(textures) - 1

    ts = int(texture_size)
    t_grass = add_tex(_noise_texture(rs, ts, 7, (40, 70, 25), (150, 190, 90)))
    t_wood = add_tex(_noise_texture(rs, ts, 5, (90, 50, 20), (215, 160, 95)))
    t_brick = add_tex(_noise_texture(rs, ts, 6, (120, 45, 35), (215, 120, 95), cell=128))
    t_marble = add_tex(_noise_texture(rs, ts, 8, (235, 235, 230), (110, 115, 130)))

    M = dict(default=0, grass=1, wood_phong=2, brick=3, marble_phong=4, red_phong=5, blue_lambert=6, glass=7, water=8, metal=9,
             white_lambert=10)
    mats = _records([
        material_create(),                                                     # Material_Create(Material*), :852-861
        material_create(S.MAT_STANDART, texture_id=t_grass),                   # Lambert + file texture
        material_create(S.MAT_VARNHISHED, texture_id=t_wood),                  # Phong + file texture
        material_create(S.MAT_STANDART, texture_id=t_brick),                   # Blinn -> MAT_STANDART (:910-932)
        material_create(S.MAT_VARNHISHED, texture_id=t_marble),
        material_create(S.MAT_VARNHISHED, color=(0.75, 0.12, 0.1, 0)),         # Phong, plain colour
        material_create(S.MAT_STANDART, color=(0.15, 0.3, 0.8, 0)),            # Lambert, plain colour
        material_create(S.MAT_GLASS, color=(0.92, 0.97, 1.0, 0), opacity=0.08),
        material_create(S.MAT_WATER, color=(0.55, 0.75, 0.9, 0)),
        material_create(S.MAT_METAL, color=(0.9, 0.88, 0.8, 0)),
        material_create(S.MAT_STANDART, color=(0.8, 0.8, 0.8, 0)),             # the importer's 0.8 grey for black colours (:880)
    ], S.Material)

    def ground_z(x, y):
        return (0.9 * np.sin(0.31 * x + 0.4) * np.cos(0.27 * y - 0.2) + 0.35 * np.sin(0.83 * x - 1.1) * np.sin(0.71 * y + 0.6)
                + 0.12 * np.sin(2.3 * x + 0.3 * y) - 1.1 * np.exp(-((x - 5.0) ** 2 + (y + 3.0) ** 2) / 9.0))

    parts = []
    # terrain
    n = int(det["terrain"])
    t = np.linspace(-20.0, 20.0, n + 1)
    X, Y = np.meshgrid(t, t, indexing="ij")
    Z = ground_z(X, Y)
    e = 1e-3
    gx, gy = (ground_z(X + e, Y) - ground_z(X - e, Y)) / (2 * e), (ground_z(X, Y + e) - ground_z(X, Y - e)) / (2 * e)
    nrm = np.stack([-gx, -gy, np.ones_like(gx)], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    uv = np.stack([(X + 20.0) / 40.0 * 6.0, (Y + 20.0) / 40.0 * 6.0], axis=-1)
    parts.append(_grid_mesh(np.stack([X, Y, Z], axis=-1), nrm, uv, M["grass"], M["default"], uv_night_scale=3.0))
    # the pond: ripples over the hollow the terrain has at (5, -3); seen from above and from below
    n = int(det["pond"])
    t = np.linspace(-3.3, 3.3, n + 1)
    X, Y = np.meshgrid(t + 5.0, t - 3.0, indexing="ij")
    wave = lambda x, y: -0.55 + 0.02 * np.sin(3.1 * x) * np.cos(2.7 * y)
    Z = wave(X, Y)
    gx, gy = (wave(X + e, Y) - wave(X - e, Y)) / (2 * e), (wave(X, Y + e) - wave(X, Y - e)) / (2 * e)
    nrm = np.stack([-gx, -gy, np.ones_like(gx)], axis=-1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    parts.append(_grid_mesh(np.stack([X, Y, Z], axis=-1), nrm, np.stack([X, Y], axis=-1) * 0.1, M["water"], M["water"]))
    # six spheres standing on the terrain
    n = int(det["sphere"])
    for (x, y, r), (mp, mn, tiles) in zip([(-3.0, 2.0, 1.5), (0.5, -1.0, 1.2), (3.5, 3.5, 1.0), (-6.0, -3.0, 1.3), (-1.0, 6.5, 1.6), (8.0, 1.5, 1.1)],
                                          [(M["wood_phong"], None, 2.0), (M["glass"], None, 1.0), (M["metal"], None, 1.0), (M["brick"], None, 3.0),
                                           (M["red_phong"], M["white_lambert"], 1.0), (M["blue_lambert"], None, 1.0)]):
        parts += _cube_sphere((x, y, float(ground_z(x, y)) + r * 0.96), r, n, mp, mn, uv_tiles=tiles)
    # three brick walls around the set (the fourth side is where the camera stands): what is lit by the sky also lights its
    # neighbours, so paths last - a modelled set is rarely a lone object under an open sky
    nw, nh = det.get("wall", (8, 4))
    for (x0, y0, x1, y1) in ((-12.0, 12.0, 12.0, 12.0), (-12.0, -14.0, -12.0, 12.0), (12.0, 12.0, 12.0, -14.0)):
        sx, sz = np.meshgrid(np.linspace(0.0, 1.0, nw + 1), np.linspace(-2.5, 4.5, nh + 1), indexing="ij")
        X, Y = x0 + (x1 - x0) * sx, y0 + (y1 - y0) * sx
        bump = 0.03 * np.sin(37.0 * sx * (abs(x1 - x0) + abs(y1 - y0))) * np.sin(5.0 * sz)   # (not a plane: a wall that was sculpted)
        nx, ny = (y1 - y0), -(x1 - x0)
        ln = np.hypot(nx, ny)
        nx, ny = nx / ln, ny / ln
        P = np.stack([X + nx * bump, Y + ny * bump, sz], axis=-1)
        Nw = np.broadcast_to(np.array([nx, ny, 0.0]), P.shape)
        uvw = np.stack([sx * 6.0, (sz + 2.5) / 7.0 * 2.0], axis=-1)
        parts.append(_grid_mesh(P, Nw, uvw, M["brick"], M["white_lambert"], uv_night_scale=1.0))
    # a marble torus lying on the slope
    nu, nv = det["torus"]
    a, b = np.meshgrid(np.linspace(0, 2 * np.pi, nu + 1), np.linspace(0, 2 * np.pi, nv + 1), indexing="ij")
    a[-1], b[:, -1] = a[0], b[:, 0]  # (closed exactly: the seam's vertices are the same numbers)
    R, r0 = 2.2, 0.55
    cx, cy = -7.5, 4.0
    P = np.stack([cx + (R + r0 * np.cos(b)) * np.cos(a), cy + (R + r0 * np.cos(b)) * np.sin(a), float(ground_z(cx, cy)) + 0.75 + r0 * np.sin(b)], axis=-1)
    Nn = np.stack([np.cos(b) * np.cos(a), np.cos(b) * np.sin(a), np.sin(b)], axis=-1)
    au, bv = np.meshgrid(np.linspace(0, 4, nu + 1), np.linspace(0, 1, nv + 1), indexing="ij")
    parts.append(_grid_mesh(P, Nn, np.stack([au, bv], axis=-1), M["marble_phong"], None, uv_night_scale=1.0))
    tris = _concat_tris(parts)

    lights = _records([
        light_point((2.0, -6.0, 7.0), color=(1, 0.95, 0.88, 1), power=55.0),
        light_spot((-9.0, -5.0, 9.0), (0.6, 0.55, -1.0), cone_angle=0.9, penumbra_angle=0.25, color=(0.9, 0.92, 1, 1), intensity=2.5),
        light_directional((-0.35, 0.45, -1.0), color=(1, 0.98, 0.9, 1), power=0.7),
    ], S.Light)
    eye = np.array([1.5, -17.0, 5.5])
    view = np.array([-0.06, 1.0, -0.24])
    view /= np.linalg.norm(view)
    right = np.cross(view, [0, 0, 1.0])
    right /= np.linalg.norm(right)
    up = np.cross(right, view)
    # Maya's default film back: 36 x 24 mm (1.417 x 0.945 in), 35 mm lens; the vertical aperture follows the image's aspect
    pos, d, r, u = camera_from_film(eye, view, up, right, 35.0, 1.417, 1.417 * height / width)
    return Scene(tris, lights, mats, np.array(textures, dtype=S.Texture), np.concatenate(texels), sky, pos, d, r, u,
                 name="mayalike" if detail is None else "mayalike_s")


FEATURES = ("plain", "glass", "water", "varnish", "varnish_textured", "textured", "metal", "two_sided", "fallback_normals",
            "spot", "directional", "cubemap")


def feature_scene(feature, width, height):
    """One kernel feature at a time on the same small stage (floor, back wall, one smooth sphere, one light, one
    specialisation of the reference kernel: 1 light): the parity tests use these to tell WHICH branch of the integrator a
    difference against the reference comes from.  ``plain`` is the control (diffuse colours, point light, black sky)."""
    if feature not in FEATURES:
        raise ValueError(f"unknown feature {feature!r}")
    texels = []
    textures = []

    def add_tex(img, w, h):
        off = sum(len(t) for t in texels)
        texels.append(img)
        textures.append((w, h, off))
        return len(textures) - 1

    if feature == "cubemap":
        sky = np.zeros((), dtype=S.Sky)
        sky["cosRotationAngle"] = f32(np.cos(0.7))
        sky["sinRotationAngle"] = f32(np.sin(0.7))
        sky["groundScale"] = 1
        cols = [((40, 40, 60), (90, 90, 140)), ((200, 120, 60), (250, 220, 160)), ((60, 140, 200), (160, 220, 250)),
                ((120, 200, 120), (220, 250, 220)), ((200, 200, 80), (250, 250, 200)), ((120, 160, 250), (230, 240, 255))]
        for i, (c0, c1) in enumerate(cols):
            off = sum(len(t) for t in texels)
            texels.append(_gradient(16, 16, c0, c1))
            sky["skyTextures"][i] = (16, 16, off)
    else:
        sky, t0 = no_sky((0, 0, 0, 0))
        texels.append(t0)

    FLOOR, WALL, BALL, BACK = 0, 1, 2, 3
    mats = [material_create(color=(0.75, 0.75, 0.7, 0)), material_create(color=(0.3, 0.45, 0.8, 0)),
            material_create(color=(0.8, 0.25, 0.2, 0)), material_create(color=(0.2, 0.7, 0.3, 0))]
    if feature == "glass":
        mats[BALL] = material_create(S.MAT_GLASS, color=(0.9, 0.95, 1.0, 0), opacity=0.1)
    elif feature == "water":
        mats[BALL] = material_create(S.MAT_WATER, color=(0.5, 0.7, 0.9, 0))
    elif feature == "varnish":
        mats[BALL] = material_create(S.MAT_VARNHISHED, color=(0.2, 0.6, 0.3, 0))
    elif feature == "varnish_textured":
        mats[BALL] = material_create(S.MAT_VARNHISHED, texture_id=add_tex(_gradient(24, 8, (150, 90, 40), (220, 170, 90)), 24, 8))
    elif feature == "textured":
        mats[FLOOR] = material_create(S.MAT_STANDART, texture_id=add_tex(_checker(32, 32, (230, 230, 230, 0), (40, 40, 160, 0)), 32, 32))
        mats[BALL] = material_create(S.MAT_STANDART, texture_id=add_tex(_gradient(24, 8, (150, 90, 40), (220, 170, 90)), 24, 8))
    elif feature == "metal":
        mats[BALL] = material_create(S.MAT_METAL, color=(0.9, 0.9, 0.9, 0))

    parts = []
    g = 4
    xs = np.linspace(-5, 5, g + 1)
    for i in range(g):
        for j in range(g):
            a, b, c, d = (xs[i], xs[j], 0), (xs[i + 1], xs[j], 0), (xs[i + 1], xs[j + 1], 0), (xs[i], xs[j + 1], 0)
            s1, s2, s3 = _quad(a, b, c, d)
            uv = np.array([[(i, j), (i + 1, j), (i + 1, j + 1)], [(i, j), (i + 1, j + 1), (i, j + 1)]], f32) * f32(0.5)
            parts.append(triangle_create(s1, s2, s3, uvp=uv, uvn=uv, mat_pos=FLOOR, mat_neg=FLOOR))
    s1, s2, s3 = _quad((-5, 5, 0), (5, 5, 0), (5, 5, 4), (-5, 5, 4))
    wall_normals = np.zeros((2, 3, 3), f32) if feature == "fallback_normals" else None  # zero -> N, which carries w = 1
    parts.append(triangle_create(s1, s2, s3, normals=wall_normals, mat_pos=WALL, mat_neg=BACK if feature == "two_sided" else WALL))
    parts.append(_sphere_tris((0.0, 1.0, 1.2), 1.2, 2, BALL, mat_neg=BACK if feature == "two_sided" else None))
    if feature == "two_sided":  # a free-standing sheet seen from both sides
        s1, s2, s3 = _quad((-4, -1, 0), (-2, 1, 0), (-2, 1, 2.5), (-4, -1, 2.5))
        parts.append(triangle_create(s1, s2, s3, mat_pos=BALL, mat_neg=BACK))
    tris = _concat_tris(parts)

    if feature == "spot":
        light = light_spot((-3.0, -4.0, 6.0), (0.45, 0.7, -1.0), cone_angle=0.8, penumbra_angle=0.3, color=(0.9, 0.9, 1, 1), intensity=12.0)
    elif feature == "directional":
        light = light_directional((-0.3, 0.4, -1.0), color=(1, 1, 0.9, 1), power=1.2)
    else:
        light = light_point((3.0, -4.0, 6.0), color=(1, 0.95, 0.9, 1), power=60.0)
    lights = _records([light], S.Light)
    span = 0.9
    pos, d, r, u = camera((0.5, -10.0, 3.0), (0, 1, -0.2), (span, 0, 0), (0, 0.2 * span * height / width, span * height / width))
    return Scene(tris, lights, _records(mats, S.Material), np.array(textures, dtype=S.Texture) if textures else np.zeros(0, S.Texture),
                 np.concatenate(texels), sky, pos, d, r, u, name="feat_" + feature)


def fuzz_scene(seed, width, height, n_lights=1, hostile=False):
    """A scene drawn from a seed: nothing in it is arranged to please the integrator.  For the parity tests that run the
    reference kernel beside the integrator (tests/test_reference_default_gpu.py): whatever the importer's conventions
    (``triangle_create`` ...) can produce, in proportions no hand-made scene has.

    * 40-300 triangles (every fourth seed: clusters of a few thousand) at scales from 0.02 to 8 around the origin inside a
      (sometimes missing) box of six big quads: free soup, fans sharing a vertex, slivers, stacks of coplanar and of exactly
      coincident triangles (distance ties; every fifth seed up to 40 of them: a leaf that cannot be split), axis-aligned
      sheets (flat bounding boxes), one or two smooth spheres;
    * per-vertex normals: flat, smooth, zero (the importer's fallback to N with w = 1), or tilted far off the face;
    * 3-9 materials of all five types with colours from 0 to 1.2, opacities from 0 to 1, a third of them textured
      (odd sizes down to 1x1; uv from -3 to 4, so every wrap is taken), different materials on the two sides;
    * ``n_lights`` lights of random types, some inside geometry, some far away, power over four decades
      (the reference bakes the light count into its program: 1 and 3 have code objects under oracle/_ref);
    * a cube-map sky with random face sizes, rotation, ground scale and exposure factors - or none;
    * a camera somewhere around, looking near the origin, with an arbitrary (not orthonormal) film frame.
    ``hostile`` adds what a mesh can hold and the file format carries: zero-area triangles (one point three times, two equal
    vertices, three collinear ones: N is 0/0, and the reference's test then ACCEPTS them for every ray that reaches them -
    all its rejections are comparisons, false for a NaN), vertices 1e6 away, a tiny triangle, a light exactly on a vertex.
    """
    rs = np.random.RandomState(1000003 * int(seed) + 17 * int(n_lights) + (7 if hostile else 0))  # (any hostile set: the same base scene)
    U = lambda lo, hi, *shape: rs.uniform(lo, hi, shape).astype(f32) if shape else f32(rs.uniform(lo, hi))
    texels, textures = [], []

    def add_tex(w, h):
        img = rs.randint(0, 256, (w * h, 4)).astype(np.uint8)
        off = sum(len(t) for t in texels)
        texels.append(img)
        textures.append((w, h, off))
        return len(textures) - 1

    if rs.rand() < 0.6:
        sky = np.zeros((), dtype=S.Sky)
        a = rs.uniform(0, 2 * np.pi)
        sky["cosRotationAngle"], sky["sinRotationAngle"] = f32(np.cos(a)), f32(np.sin(a))
        sky["groundScale"] = U(0.3, 3.0)
        sky["exposantFactorX"], sky["exposantFactorY"] = (U(0.0, 2.0), U(0.0, 2.0)) if rs.rand() < 0.5 else (f32(0), f32(0))
        for i in range(6):
            w, h = int(rs.choice([1, 2, 5, 16, 31])), int(rs.choice([1, 3, 8, 16]))
            off = sum(len(t) for t in texels)
            texels.append(rs.randint(0, 256, (w * h, 4)).astype(np.uint8))
            sky["skyTextures"][i] = (w, h, off)
    else:
        sky, t0 = no_sky(tuple(int(v) for v in rs.randint(0, 256, 4)))
        texels.append(t0)

    n_mat = int(rs.randint(3, 10))
    mats = []
    for _ in range(n_mat):
        mtype = int(rs.choice([S.MAT_STANDART, S.MAT_STANDART, S.MAT_GLASS, S.MAT_WATER, S.MAT_VARNHISHED, S.MAT_METAL]))
        tex = add_tex(int(rs.choice([1, 2, 3, 7, 17, 32])), int(rs.choice([1, 2, 5, 8, 29]))) if rs.rand() < 0.35 else -1
        mats.append(material_create(mtype, color=tuple(U(0.0, 1.2, 3)) + (0.0,), texture_id=tex, opacity=float(U(0.0, 1.0))))
    pick = lambda: int(rs.randint(0, n_mat))

    parts = []

    def add(s, normals_mode=None, n=None):
        """s: (k,3,3) vertices"""
        s = np.asarray(s, f32)
        k = len(s)
        mode = normals_mode if normals_mode is not None else rs.choice(["flat", "zero", "tilted", "flat"])
        if mode == "flat":
            nrm = None
        elif mode == "zero":
            nrm = np.zeros((k, 3, 3), f32)
        elif mode == "given":
            nrm = n
        else:
            c = np.cross(s[:, 1] - s[:, 0], s[:, 2] - s[:, 0])
            nrm = (c[:, None, :] / np.maximum(np.linalg.norm(c, axis=-1), 1e-30)[:, None, None] + rs.uniform(-0.9, 0.9, (k, 3, 3))).astype(f32)
        uvp = U(-3.0, 4.0, k, 3, 2)
        uvn = U(-3.0, 4.0, k, 3, 2) if rs.rand() < 0.5 else uvp
        parts.append(triangle_create(s[:, 0], s[:, 1], s[:, 2], normals=nrm, uvp=uvp, uvn=uvn, mat_pos=pick(),
                                     mat_neg=pick() if rs.rand() < 0.5 else None))

    if rs.rand() < 0.7:  # the room
        L = float(U(4.0, 9.0))
        c = [(-L, -L, -L), (L, -L, -L), (L, L, -L), (-L, L, -L), (-L, -L, L), (L, -L, L), (L, L, L), (-L, L, L)]
        for a, b, cc, d in ((0, 1, 2, 3), (4, 7, 6, 5), (0, 4, 5, 1), (3, 2, 6, 7), (0, 3, 7, 4), (1, 5, 6, 2)):
            if rs.rand() < 0.85:
                s1, s2, s3 = _quad(c[a], c[b], c[cc], c[d])
                add(np.stack([s1, s2, s3], axis=1), normals_mode="flat")
    for _ in range(int(rs.randint(8, 60))):  # soup at mixed scales
        scale = float(10 ** rs.uniform(-1.7, 0.9))
        ctr = U(-4.0, 4.0, 3)
        add((ctr + U(-scale, scale, 1, 3, 3)).astype(f32))
    if seed % 4 == 3:  # every fourth scene: a few thousand small triangles in clusters (a tree 12-16 levels deep)
        for _ in range(int(rs.randint(3, 9))):
            k, ctr, spread = int(rs.randint(100, 700)), U(-3.5, 3.5, 3), float(10 ** rs.uniform(-0.8, 0.3))
            c = ctr + rs.normal(0, spread, (k, 1, 3)).astype(f32)
            add((c + U(-0.08, 0.08, k, 3, 3) * f32(spread * 2)).astype(f32), normals_mode=rs.choice(["flat", "tilted"]))
    if seed % 5 == 2:  # a leaf the builder cannot split: many coincident triangles (more than a leaf's count field holds)
        t = (U(-2.0, 2.0, 3) + U(-1.5, 1.5, 3, 3)).astype(f32)
        add(np.repeat(t[None], int(rs.randint(9, 40)), axis=0))
    for _ in range(int(rs.randint(1, 4))):  # fans around one vertex
        apex = U(-3.0, 3.0, 3)
        rim = apex + U(-1.5, 1.5, int(rs.randint(4, 9)), 3)
        add(np.stack([np.broadcast_to(apex, rim.shape), rim, np.roll(rim, 1, axis=0)], axis=1))
    for _ in range(int(rs.randint(1, 4))):  # slivers
        a = U(-4.0, 4.0, 3)
        b = a + U(-3.0, 3.0, 3)
        add(np.stack([a, b, (a + b) * f32(0.5) + U(-0.003, 0.003, 3)])[None])
    for _ in range(int(rs.randint(1, 4))):  # coincident copies and coplanar stacks: exact distance ties, in one leaf or in several
        t = (U(-3.0, 3.0, 3) + U(-1.2, 1.2, 3, 3)).astype(f32)
        copies = int(rs.randint(2, 5))
        for i in range(copies):
            add(t[None] if rs.rand() < 0.6 else (t + (t[1] - t[0]) * f32(0.25 * i))[None])
    for _ in range(int(rs.randint(1, 4))):  # axis-aligned sheets: bounding boxes of zero thickness
        axis = int(rs.randint(0, 3))
        q = U(-3.0, 3.0, 4, 3)
        q[:, axis] = f32(rs.choice([0.0, 1.0, -2.5, float(U(-3, 3))]))
        add(np.stack([q[[0, 0]], q[[1, 2]], q[[2, 3]]], axis=1))
    for _ in range(int(rs.randint(0, 3))):  # smooth spheres
        v, f = _icosphere(int(rs.randint(0, 3)))
        r, ctr = float(U(0.3, 1.6)), U(-3.0, 3.0, 3)
        add((v * r + ctr).astype(f32)[f], normals_mode="given", n=v.astype(f32)[f])
    hostile = {"point", "pair", "collinear", "far", "tiny", "light"} if hostile is True else set(hostile or ())
    if hostile:
        p, q = U(-2.0, 2.0, 3), U(-2.0, 2.0, 3)
        far, tiny = U(-1.0, 1.0, 1, 3, 3) * f32(1e6), U(-2.0, 2.0, 3) + U(-1e-6, 1e-6, 1, 3, 3)
        if "point" in hostile:
            add(np.stack([p, p, p])[None], normals_mode="flat")   # a point: N = 0/0
        if "pair" in hostile:
            add(np.stack([p, p, q])[None], normals_mode="zero")   # two equal vertices
        if "collinear" in hostile:  # three distinct vertices on a line: passes the importer's ordering ASSERT, N = 0/0 all the same
            a, e = np.round(p * 4) / 4, np.array([0.5, -0.25, 0.75], f32)
            add(np.stack([a, a + e, a + 2 * e]).astype(f32)[None], normals_mode="flat")
        if "far" in hostile:
            add(far.astype(f32), normals_mode="flat")             # far, huge
        if "tiny" in hostile:
            add(tiny.astype(f32), normals_mode="tilted")
    tris = _concat_tris(parts)

    lights = []
    for i in range(n_lights):
        kind = rs.choice(["point", "spot", "directional"]) if n_lights > 1 or rs.rand() < 0.6 else "point"
        col = tuple(U(0.2, 1.0, 3)) + (1.0,)
        pos = U(-6.0, 6.0, 3) * f32(1.0 if rs.rand() < 0.8 else 20.0)
        if "light" in hostile and i == 0:
            pos = tris["S1"][0][:3].copy()  # exactly on a vertex
        if kind == "point":
            lights.append(light_point(pos, color=col, power=float(10 ** rs.uniform(-1, 3))))
        elif kind == "spot":
            lights.append(light_spot(pos, -pos + U(-2.0, 2.0, 3), cone_angle=float(U(0.2, 2.0)), penumbra_angle=float(U(0.0, 0.6)),
                                     color=col, intensity=float(10 ** rs.uniform(-1, 2))))
        else:
            lights.append(light_directional(U(-1.0, 1.0, 3) + f32(1e-3), color=col, power=float(10 ** rs.uniform(-1.5, 0.7))))
    eye = U(-7.0, 7.0, 3) * f32(rs.choice([1.0, 1.0, 0.3, 3.0]))  # (0.3: usually inside the clutter, 3: outside the room)
    view = (U(-1.0, 1.0, 3) - eye).astype(f32) * f32(rs.choice([1.0, 1.0, 1.0, -1.0]))  # (-1: looking away, mostly sky)
    view = view / np.linalg.norm(view)
    right = np.cross(view, U(-1.0, 1.0, 3))
    right = right / np.linalg.norm(right) * float(U(0.4, 1.4))
    up = np.cross(right, view) * float(U(0.5, 1.5)) * height / width + U(-0.1, 0.1, 3)
    pos, d, r, u = camera(eye, view, right, up)
    return Scene(tris, _records(lights, S.Light), _records(mats, S.Material),
                 np.array(textures, dtype=S.Texture) if textures else np.zeros(0, S.Texture), np.concatenate(texels), sky, pos, d, r, u,
                 name=f"fuzz{seed}" + ("h" if hostile else "") + f"_l{n_lights}", meta={"seed": int(seed)})


def add_zero_area_triangles(scene, k, seed=5):
    """`k` zero-area triangles as the importer emits them for a degenerate face of a mesh (three collinear vertices pass its
    ordering ASSERT; N = normalize(0) = 0/0, MayaImporter.cpp:943-1047) scattered through the scene's volume, appended to a COPY of
    the scene (no tree: call bvh_create).  Coordinates with few mantissa bits, so that a, a + e, a + 2e are EXACTLY collinear in
    float and the cross product is exactly zero."""
    import copy
    import warnings
    rs = np.random.RandomState(seed)
    lo, hi = scene.triangulation["S1"][:, :3].min(0), scene.triangulation["S1"][:, :3].max(0)
    a = (np.round(rs.uniform(lo * 0.8, hi * 0.8, (k, 3)) * 16) / 16).astype(f32)
    e = (np.round(rs.uniform(0.01, 0.06, (k, 3)) * 256) / 256).astype(f32)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        bad = triangle_create(a, a + e, a + 2 * e, mat_pos=0)
    out = copy.copy(scene)
    out.triangulation = _concat_tris([scene.triangulation, bad])
    out.bvh = None
    out.name = scene.name + f"+{k}za"
    return out


def corrupt_records(scene, seed):
    """What the arrays can hold although no importer writes it - the integrator takes raw records (the scene-cache files,
    any host that fills GlobalVars itself), and whatever the reference makes of them is the contract: geometric normals that
    are not unit length, flipped, or belong to another plane; vertices out of the importer's lexicographic order; w
    components that are not 1 (all three equal, or each its own: the triangle test's generic form); vertex normals of any
    length; material and light types outside their enums, opacities outside [0, 1], negative powers; a spot light whose
    inner cone is the narrower one.  In place, before the tree is built; returns the scene."""
    rs = np.random.RandomState(77003 * int(seed) + 5)
    t = scene.triangulation
    n = len(t)
    pickn = lambda frac: np.flatnonzero(rs.rand(n) < frac)
    for i in pickn(0.25):
        t["N"][i][:3] *= np.float32(rs.choice([-1.0, 0.3, 2.5, 1.0 + 1e-3]))
    for i in pickn(0.05):
        t["N"][i][:3] = (t["N"][i][:3] + rs.uniform(-0.5, 0.5, 3)).astype(f32)
    for i in pickn(0.1):
        a, b = rs.choice(3, 2, replace=False)
        for grp in (("S1", "S2", "S3"), ("N1", "N2", "N3"), ("UVP1", "UVP2", "UVP3"), ("UVN1", "UVN2", "UVN3")):
            x, y = t[grp[a]][i].copy(), t[grp[b]][i].copy()
            t[grp[a]][i], t[grp[b]][i] = y, x
    mode = rs.choice(["importer", "equal", "own"])
    if mode != "importer":
        for i in pickn(0.5 if mode == "equal" else 0.2):
            w3 = rs.uniform(-2, 3, 3).astype(f32) if mode == "own" else np.repeat(f32(rs.uniform(-2, 3)), 3)
            for k, name in enumerate(("S1", "S2", "S3")):
                t[name][i][3] = w3[k]
    for i in pickn(0.1):
        for name in ("N1", "N2", "N3"):
            t[name][i] = (t[name][i] * np.float32(rs.choice([0.0, 0.5, 3.0, -1.0]))).astype(f32)
    m = scene.materiaux
    for i in range(len(m)):
        r = rs.rand()
        if r < 0.15:
            m["type"][i] = int(rs.choice([5, 7, 255]))
        elif r < 0.3:
            m["opacity"][i] = np.float32(rs.choice([-0.5, 1.5, 0.0]))
    li = scene.lights
    for i in range(len(li)):
        r = rs.rand()
        if r < 0.2 and len(li) > 1:
            li["type"][i] = int(rs.choice([3, 9]))
        elif r < 0.35:
            li["power"][i] = -li["power"][i]
        elif r < 0.5:
            li["cosOfInnerFallOffAngle"][i], li["cosOfOuterFallOffAngle"][i] = li["cosOfOuterFallOffAngle"][i], li["cosOfInnerFallOffAngle"][i]
        if rs.rand() < 0.3:
            li["direction"][i][:3] *= np.float32(rs.choice([0.0, 2.0]))  # (a direction that is not a unit vector)
    scene.name += "r"
    return scene


def corrupt_tree(scene, seed):
    """The tree is the caller's too (the reference builds it on the host; the integrator takes it as an array): boxes that
    do not bound what is below them (shrunk, grown), inverted on an axis (pMin > pMax), with a NaN or an infinite face,
    marked empty; split axes that are not the builder's; leaves that hold fewer triangles than the builder gave them, or
    none.  Structure (child indices, leaf ranges inside the triangle array) stays valid - the integrator refuses a tree that
    would make it read out of bounds, which the reference would simply do.  In place, after the tree is built."""
    rs = np.random.RandomState(990001 * int(seed) + 3)
    b = scene.bvh
    box = b["trianglesAABB"]
    for i in range(1, len(b)):
        r = rs.rand()
        if r < 0.15:
            c = (box["pMin"][i] + box["pMax"][i]) * f32(0.5)
            k = rs.choice([0.3, 0.8, 1.5], 4).astype(f32)
            half = (box["pMax"][i] - box["pMin"][i]) * f32(0.5) * k
            box["pMin"][i], box["pMax"][i] = c - half, c + half
        elif r < 0.18:
            a = int(rs.randint(0, 3))
            box["pMin"][i][a], box["pMax"][i][a] = box["pMax"][i][a], box["pMin"][i][a]
        elif r < 0.20:
            box[rs.choice(["pMin", "pMax"])][i][int(rs.randint(0, 3))] = np.nan
        elif r < 0.22:
            box["pMax"][i][int(rs.randint(0, 3))] = np.inf
        elif r < 0.23:
            box["pMin"][i][int(rs.randint(0, 3))] = -np.inf
        elif r < 0.26:
            box["isEmpty"][i] = 1
        if b["isLeaf"][i]:
            if rs.rand() < 0.05 and b["nbTriangles"][i] > 0:
                b["nbTriangles"][i] -= 1
        elif rs.rand() < 0.1:
            b["cutAxis"][i] = int(rs.randint(0, 3))
    scene.name += "t"
    return scene


def build(name, width, height):
    """Named scenes used by tests, fixtures and the bench."""
    if name == "cornell":
        return cornell_box(width, height)
    if name == "matmix":
        return material_mix(width, height)
    if name == "mayalike":
        return maya_like(width, height)
    if name == "mayalike_s":
        return maya_like(width, height, detail=MAYALIKE_SMALL)
    if name.startswith("feat_"):
        return feature_scene(name[5:], width, height)
    m = re.fullmatch(r"fuzz(\d+)(h?)(r?)_l(\d+)", name)
    if m:
        sc = fuzz_scene(int(m.group(1)), width, height, n_lights=int(m.group(4)), hostile=bool(m.group(2)))
        return corrupt_records(sc, int(m.group(1))) if m.group(3) else sc
    if name.startswith("tris"):
        spec = name[4:]
        n = int(spec[:-1]) * {"k": 1000, "m": 1000000}[spec[-1]] if spec[-1] in "km" else int(spec)
        return random_triangles(n, width, height)
    raise ValueError(f"unknown scene {name!r}")
