"""ctypes bindings of the CHECKERS (test infrastructure only):

* oracle/build/libpt_oracle.so   - the CPU restatement of the reference integrator
* oracle/_ref/libref_bvh.so      - the reference's own BVH_Create, compiled unmodified (x86-64)
* oracle/_ref/libref_gpu_runner.so + ref_kernel_*.hsaco - the reference's own Kernel_Main on the GPU

Nothing under opencl_pathtracer_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from opencl_pathtracer_amd import structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "build", "libpt_oracle.so")         # the reference's strict build's arithmetic
ORACLE_DA_LIB = os.path.join(ORACLE_DIR, "build", "libpt_oracle_da.so")  # ... its default build's (what its own build line gives)
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
REF_BVH_LIB = os.path.join(REF_DIR, "libref_bvh.so")
REF_GPU_LIB = os.path.join(REF_DIR, "libref_gpu_runner.so")


class F4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class PtoScene(C.Structure):
    _fields_ = [("bvh", C.c_void_p), ("triangulation", C.c_void_p), ("lights", C.c_void_p), ("materiaux", C.c_void_p),
                ("textures", C.c_void_p), ("textures_data", C.c_void_p), ("sky", C.c_void_p),
                ("camera_position", F4), ("camera_direction", F4), ("camera_right", F4), ("camera_up", F4),
                ("image_width", C.c_uint32), ("image_height", C.c_uint32), ("ray_max_depth", C.c_uint32),
                ("lights_size", C.c_uint32), ("sampler", C.c_uint32), ("super_sampling", C.c_uint32),
                ("x2inv", C.c_void_p), ("russian_roulette", C.c_uint32), ("source_seed", C.c_uint32),
                ("first_sample_guard", C.c_uint32)]


class PtoBuffers(C.Structure):
    _fields_ = [("image_color", C.c_void_p), ("image_ray_nb", C.c_void_p), ("image_v", C.c_void_p),
                ("ray_depths", C.c_void_p), ("ray_intersected_bbx", C.c_void_p), ("ray_intersected_tri", C.c_void_p)]


class PtoTotals(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "segments", "surface_hits", "shadow_rays", "box_tests",
                                         "triangle_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class PtoBounce(C.Structure):
    _fields_ = [("triangle_id", C.c_uint32), ("material_id", C.c_uint32), ("s", C.c_float), ("t", C.c_float),
                ("point", C.c_float * 4), ("ns", C.c_float * 4), ("out_dir", C.c_float * 4),
                ("transfer", C.c_float * 4), ("radiance", C.c_float * 4), ("seed_after", C.c_int32),
                ("n_bbx", C.c_uint32), ("n_tri", C.c_uint32)]


_oracle = {}
_hw_tables = {}  # kept alive: the oracle holds pointers into them


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


def _hw_table(kind, entries):
    """2-bit deviations of a gfx950 instruction from the correctly rounded function (tests/golden/make_hw_tables.py)"""
    if kind not in _hw_tables:
        t = np.ascontiguousarray(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"{kind}_gfx950.npz"))["packed"])
        assert t.dtype == np.uint8 and t.size == entries // 4
        _hw_tables[kind] = t
    return _hw_tables[kind].ctypes.data_as(C.c_void_p)


def oracle(default_arithmetic=False):
    """The CPU restatement in the arithmetic of the reference's strict build, or (default_arithmetic) of the build its own
    build line produces."""
    path = ORACLE_DA_LIB if default_arithmetic else ORACLE_LIB
    if default_arithmetic not in _oracle:
        if not os.path.exists(path):
            build_oracle()
        lib = C.CDLL(path)
        assert lib.pto_default_arithmetic() == (1 if default_arithmetic else 0)
        lib.pto_render.argtypes = [C.POINTER(PtoScene), C.c_uint32, C.c_uint32, C.POINTER(PtoBuffers), C.c_int,
                                   C.POINTER(PtoTotals)]
        lib.pto_render.restype = None
        lib.pto_trace_path.argtypes = [C.POINTER(PtoScene), C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(PtoBounce),
                                       C.c_int, C.POINTER(C.c_float)]
        lib.pto_random.argtypes = [C.POINTER(C.c_int32)]
        lib.pto_random.restype = C.c_float
        lib.pto_initialize_random_seed.argtypes = [C.c_uint32] * 5
        lib.pto_initialize_random_seed.restype = C.c_int32
        lib.pto_sampler.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        lib.pto_sampler.restype = None
        lib.pto_sincos.argtypes = [C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_sincos.restype = None
        lib.pto_concentric_sample_disk.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_concentric_sample_disk.restype = None
        lib.pto_bounding_box_intersects.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float]
        lib.pto_triangle_intersects.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                C.POINTER(C.c_float)]
        lib.pto_fresnel_glass.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_fresnel_glass.restype = C.c_float
        lib.pto_fresnel_varnish.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_fresnel_varnish.restype = C.c_float
        lib.pto_sky_color.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_sky_color.restype = None
        lib.pto_cosine_sample_hemisphere.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        lib.pto_cosine_sample_hemisphere.restype = None
        # normalize()'s reciprocal square root on the platform the reference runs on is a hardware instruction: the
        # oracle reproduces it from the table measured on an MI355X (tests/golden/make_rsq_table.py)
        for kind, entries in (("rsq", 1 << 24), ("rcp", 1 << 23), ("sqrt", 1 << 24)):
            setter = getattr(lib, f"pto_set_{kind}_table")
            setter.argtypes = [C.c_void_p]
            setter.restype = None
            setter(_hw_table(kind, entries))
            fn = getattr(lib, f"pto_hardware_{kind}")
            fn.argtypes = [C.c_float]
            fn.restype = C.c_float
        _oracle[default_arithmetic] = lib
    return _oracle[default_arithmetic]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _f4(v):
    v = np.asarray(v, np.float32)
    return F4(float(v[0]), float(v[1]), float(v[2]), float(v[3]))


def x2inv_table():
    """The adaptive sampler's table, computed here from its definition (chi2.ppf(0.01, n), six decimals) --
    independently of the product's generated x2inv_table.inc."""
    from scipy.stats import chi2
    t = np.concatenate([[0.0], chi2.ppf(0.01, np.arange(1, 1001))])
    return np.round(t, 6).astype(np.float32)


class OracleScene:
    """Keeps the numpy arrays alive next to the C struct that points into them."""

    def __init__(self, scene, width, height, ray_max_depth, sampler=S.JITTERED, super_sampling=False, russian_roulette=False,
                 source_seed=False, first_sample_guard=False):
        self.arrays = [np.ascontiguousarray(a) for a in (scene.bvh, scene.triangulation, scene.lights, scene.materiaux,
                                                         scene.textures, scene.texturesData, scene.sky)]
        s = PtoScene()
        (s.bvh, s.triangulation, s.lights, s.materiaux, s.textures, s.textures_data) = [_vp(a) for a in self.arrays[:6]]
        s.sky = self.arrays[6].ctypes.data_as(C.c_void_p)
        s.camera_position, s.camera_direction = _f4(scene.cameraPosition), _f4(scene.cameraDirection)
        s.camera_right, s.camera_up = _f4(scene.cameraRight), _f4(scene.cameraUp)
        s.image_width, s.image_height, s.ray_max_depth = width, height, ray_max_depth
        s.lights_size, s.sampler, s.super_sampling, s.x2inv = len(scene.lights), sampler, 0, None
        if super_sampling:
            self.x2inv = x2inv_table()
            s.super_sampling, s.x2inv = 1, self.x2inv.ctypes.data_as(C.c_void_p)
        s.russian_roulette = 1 if russian_roulette else 0
        s.source_seed = 1 if source_seed else 0
        s.first_sample_guard = 1 if first_sample_guard else 0
        self.c = s
        self.width, self.height, self.depth = width, height, ray_max_depth


def oracle_render(scene, width, height, ray_max_depth, n_iterations, first_iteration=0, sampler=S.JITTERED,
                  n_threads=8, into=None, super_sampling=False, image_v=None, russian_roulette=False, default_arithmetic=False,
                  source_seed=False, first_sample_guard=False):
    """Returns (imageColor[H,W,4], imageRayNb[H,W], (depths, bbx, tri), totals dict).  `into` = a previous
    result tuple to keep accumulating into (iteration ranges must then be rendered in order)."""
    lib = oracle(default_arithmetic)
    osc = OracleScene(scene, width, height, ray_max_depth, sampler, super_sampling, russian_roulette, source_seed, first_sample_guard)
    if into is None:
        color = np.zeros((height, width, 4), np.float32)
        count = np.zeros((height, width), np.float32)
        depths = np.zeros(ray_max_depth + 1, np.uint32)
        bbx = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
        tri = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
    else:
        color, count, (depths, bbx, tri), _ = into
    imgv = np.zeros((height, width, 4), np.float32) if image_v is None else image_v
    buf = PtoBuffers(_vp(color), _vp(count), _vp(imgv), _vp(depths), _vp(bbx), _vp(tri))
    tot = PtoTotals()
    lib.pto_render(C.byref(osc.c), first_iteration, n_iterations, C.byref(buf), n_threads, C.byref(tot))
    return color, count, (depths, bbx, tri), tot.as_dict()


def oracle_trace(scene, width, height, ray_max_depth, x, y, iteration, sampler=S.JITTERED, default_arithmetic=False):
    lib = oracle(default_arithmetic)
    osc = OracleScene(scene, width, height, ray_max_depth, sampler)
    bounces = (PtoBounce * 64)()
    rad = (C.c_float * 4)()
    n = lib.pto_trace_path(C.byref(osc.c), x, y, iteration, bounces, 64, rad)
    return [bounces[i] for i in range(max(n, 0))], np.array(rad[:], np.float32)


# ----------------------------------------------------------------------------- the reference itself

def have_ref_bvh():
    return os.path.exists(REF_BVH_LIB)


def ref_bvh_create(triangulation):
    """Runs the reference's BVH_Create (compiled unmodified) on a COPY of `triangulation`.
    Returns (nodes, reordered triangles, bvhMaxDepth)."""
    lib = C.CDLL(REF_BVH_LIB)
    lib.ref_bvh_create.argtypes = [C.c_void_p, C.c_uint, C.c_void_p, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    assert lib.ref_sizeof_node() == S.Node.itemsize and lib.ref_sizeof_triangle() == S.Triangle.itemsize
    tris = np.ascontiguousarray(triangulation).copy()
    n = len(tris)
    nodes = np.zeros(max(2 * n - 1, 1), dtype=S.Node)
    size, depth = C.c_uint(0), C.c_uint(0)
    rc = lib.ref_bvh_create(_vp(tris), n, _vp(nodes), len(nodes), C.byref(size), C.byref(depth))
    assert rc == 0
    return nodes[:size.value].copy(), tris, depth.value


class RefGpuJob(C.Structure):
    _fields_ = [("hsaco_path", C.c_char_p), ("width", C.c_uint32), ("height", C.c_uint32), ("ray_max_depth", C.c_uint32),
                ("local_x", C.c_uint32), ("local_y", C.c_uint32), ("first_iteration", C.c_uint32),
                ("n_iterations", C.c_uint32),
                ("camera_position", C.c_float * 4), ("camera_direction", C.c_float * 4), ("camera_right", C.c_float * 4),
                ("camera_up", C.c_float * 4),
                ("bvh", C.c_void_p), ("bvh_bytes", C.c_uint64), ("triangulation", C.c_void_p),
                ("triangulation_bytes", C.c_uint64), ("lights", C.c_void_p), ("lights_bytes", C.c_uint64),
                ("materiaux", C.c_void_p), ("materiaux_bytes", C.c_uint64), ("textures", C.c_void_p),
                ("textures_bytes", C.c_uint64), ("textures_data", C.c_void_p), ("textures_data_bytes", C.c_uint64),
                ("sky", C.c_void_p),
                ("image_color", C.c_void_p), ("image_ray_nb", C.c_void_p), ("ray_depths", C.c_void_p),
                ("ray_bbx", C.c_void_p), ("ray_tri", C.c_void_p), ("kernel_ms", C.c_double)]


def ref_kernel_path(config_name, strict=False):
    return os.path.join(REF_DIR, f"ref_kernel_{config_name}{'.strict' if strict else ''}.hsaco")


def have_ref_kernel(config_name, strict=False):
    return os.path.exists(REF_GPU_LIB) and os.path.exists(ref_kernel_path(config_name, strict))


# A GPU box that received the repository without the reference-derived checkers must not report green with the whole
# reference comparison gone: a missing object FAILS the test.  PTMI_ALLOW_MISSING_REFERENCE=1 (a third-party box that
# cannot hold the reference's binaries) turns that into a skip again.
ALLOW_MISSING_REFERENCE = os.environ.get("PTMI_ALLOW_MISSING_REFERENCE", "") not in ("", "0")


def missing_reference(what):
    import pytest
    if ALLOW_MISSING_REFERENCE:
        pytest.skip(what + " (PTMI_ALLOW_MISSING_REFERENCE is set)")
    pytest.fail(what + ": build it where the reference tree exists (make -C oracle ref) and ship oracle/_ref/ with the "
                "repository, or set PTMI_ALLOW_MISSING_REFERENCE=1 to skip the reference comparisons", pytrace=False)


def configured_ref_objects():
    """Every file `make -C oracle ref` leaves behind for the GPU tests: the launcher, the reference's builder and both builds
    of every specialisation named in oracle/ref_configs.txt."""
    paths = [REF_GPU_LIB, REF_BVH_LIB]
    with open(os.path.join(ORACLE_DIR, "ref_configs.txt")) as f:
        for line in f:
            line = line.split("#")[0].split()
            if line:
                paths += [ref_kernel_path(line[0]), ref_kernel_path(line[0], strict=True)]
    return paths


def _local_size(n):
    for c in (8, 4, 2, 1):
        if n % c == 0:
            return c


def ref_gpu_render(config_name, scene, width, height, ray_max_depth, n_iterations, first_iteration=0, strict=False):
    """Runs the REFERENCE kernel (code object built from its unmodified source) on the GPU.
    strict=False: OpenCL default arithmetic; strict=True: the -ffp-contract=off / correctly-rounded build.
    Returns (imageColor, imageRayNb, (depths, bbx, tri), kernel_ms)."""
    lib = C.CDLL(REF_GPU_LIB)
    lib.ref_gpu_run.argtypes = [C.POINTER(RefGpuJob)]
    lib.ref_gpu_last_error.restype = C.c_char_p
    arrs = [np.ascontiguousarray(a) for a in (scene.bvh, scene.triangulation, scene.lights, scene.materiaux,
                                               scene.textures, scene.texturesData, scene.sky)]
    color = np.zeros((height, width, 4), np.float32)
    count = np.zeros((height, width), np.float32)
    depths = np.zeros(ray_max_depth + 1, np.uint32)
    bbx = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
    tri = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
    j = RefGpuJob()
    j.hsaco_path = ref_kernel_path(config_name, strict).encode()
    j.width, j.height, j.ray_max_depth = width, height, ray_max_depth
    j.local_x, j.local_y = _local_size(width), _local_size(height)
    j.first_iteration, j.n_iterations = first_iteration, n_iterations
    for name, v in (("camera_position", scene.cameraPosition), ("camera_direction", scene.cameraDirection),
                    ("camera_right", scene.cameraRight), ("camera_up", scene.cameraUp)):
        setattr(j, name, (C.c_float * 4)(*[float(x) for x in v]))
    for name, a in zip(("bvh", "triangulation", "lights", "materiaux", "textures", "textures_data"), arrs[:6]):
        setattr(j, name, _vp(a))
        setattr(j, name + "_bytes", a.nbytes)
    j.sky = arrs[6].ctypes.data_as(C.c_void_p)
    j.image_color, j.image_ray_nb = _vp(color), _vp(count)
    j.ray_depths, j.ray_bbx, j.ray_tri = _vp(depths), _vp(bbx), _vp(tri)
    rc = lib.ref_gpu_run(C.byref(j))
    if rc:
        raise RuntimeError("reference kernel launch failed: " + lib.ref_gpu_last_error().decode())
    return color, count, (depths, bbx, tri), j.kernel_ms
