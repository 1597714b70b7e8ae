"""Host-side mirror of the reference's device backend, on top of the C ABI of ``libptmi.so``.

The reference's orchestration drives its backend with three calls
(``Controleur/PathTracer_OpenCL.h:17-19``, called from ``PathTracer.cpp:74,76,82``)::

    OpenCL_SetupContext(globalVars, sampler)
    OpenCL_InitializeMemory(globalVars)
    OpenCL_RunKernel(globalVars, UpdateWindowFunc, numImagesToRender, &t1, &t2, &t3)

``Backend`` exposes the same three steps with the same meaning and order
(`setup_context`, `initialize_memory`, `run_kernel`) plus the range form the
MI355X build adds (`render(first_iteration, n)`).  Python here is plumbing: every
pixel is computed by the HIP kernels behind ``include/ptmi.h``; if the library or a
GPU is missing the calls raise -- there is no CPU path in the product.

Errors: the reference throws ``std::runtime_error`` for any backend failure
(PathTracer_OpenCL.cpp:350-387,485); here every non-zero status raises ``PtmiError``
(a RuntimeError) carrying ``ptmi_last_error``.
"""
import ctypes as C
import os
import time

import numpy as np

from . import structs as S

# PTMI_LIBRARY lets a developer A/B another in-tree build of the same ABI (tools/build_variants.sh, tools/run_variants.sh)
_LIB_PATH = os.environ.get("PTMI_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libptmi.so")
_lib = None


class PtmiError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"ptmi error {code}: {message}")
        self.code = code


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("image_width", C.c_uint32),
                ("image_height", C.c_uint32), ("ray_max_depth", C.c_uint32), ("lights_size", C.c_uint32),
                ("sampler", C.c_uint32), ("super_sampling", C.c_uint32), ("flags", C.c_uint32),
                ("n_devices", C.c_uint32), ("devices", C.c_int32 * 16)]


class SceneDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32),
                ("bvh", C.c_void_p), ("bvh_size", C.c_uint32),
                ("triangulation", C.c_void_p), ("triangulation_size", C.c_uint32),
                ("lights", C.c_void_p), ("lights_size", C.c_uint32),
                ("materiaux", C.c_void_p), ("materiaux_size", C.c_uint32),
                ("textures", C.c_void_p), ("textures_size", C.c_uint32),
                ("textures_data", C.c_void_p), ("textures_data_size", C.c_uint32),
                ("sky", C.c_void_p),
                ("camera_position", Float4), ("camera_direction", Float4), ("camera_right", Float4),
                ("camera_up", Float4)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("paths", "segments", "surface_hits", "shadow_rays", "box_tests",
                                         "triangle_tests")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class SchedulerStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("trips_node", "lanes_node", "trips_triangle", "lanes_triangle", "trips_path",
                                         "lanes_path", "cycles_path", "cycles_loop", "leaf_item_violations", "paths_retraced", "textured_hits", "workgroup_lanes",
                                         "resident_workgroups")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


USER_SNAPSHOT_SLOTS = 64  # ptmi.h: PTMI_MAX_SNAPSHOT_SLOTS - 1 (the last slot is the library's own)
class InvariantChecks(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("sample_out_of_range", "normal_not_facing_ray", "negative_direct_radiance",
                                         "scattered_below_surface", "statistics_out_of_range", "refraction_undefined_in_reference")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


FLAG_NO_HISTOGRAMS = 1
FLAG_SCHEDULER_STATS = 4  # collect scheduler_stats() (off by default)
FLAG_MEGAKERNEL = 2  # one path per lane instead of the persistent wavefront kernel (same results)
FLAG_RUSSIAN_ROULETTE = 8  # non-parity mode: the reference's commented-out termination block (FullKernel.cl:1306-1314)
# the arithmetic of the build the reference's own build line gives (OpenCL default: fused a*b+c, v_rcp_f32 division, v_sqrt_f32);
# without it the strict arithmetic (-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt).  Both bit for bit.
FLAG_DEFAULT_ARITHMETIC = 16
FLAG_SOURCE_SEED = 32  # non-parity mode: seed 1 (as the source text reads) where the compiled reference keeps seed 0 (1 path in 65536)

# every symbol include/ptmi.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = ["ptmi_setup_context", "ptmi_initialize_memory", "ptmi_render", "ptmi_synchronize", "ptmi_read_image",
               "ptmi_write_image", "ptmi_pin_host_buffer", "ptmi_unpin_host_buffer", "ptmi_snapshot", "ptmi_render_snapshots", "ptmi_read_snapshot", "ptmi_write_variance", "ptmi_read_display", "ptmi_read_statistics", "ptmi_clear", "ptmi_release", "ptmi_get_counters", "ptmi_get_scheduler_stats", "ptmi_get_invariant_checks", "ptmi_literal_kernel_reason", "ptmi_validate_scene",
               "ptmi_kernel_time", "ptmi_reduce_path",
               "ptmi_set_stream", "ptmi_device_accumulators", "ptmi_bind_accumulators", "ptmi_read_variance",
               "ptmi_device_variance", "ptmi_last_error",
               "ptmi_abi_version", "ptmi_device_count", "ptmi_device_share", "ptmi_bvh_create"]


def library_path():
    return _LIB_PATH


def load_library():
    """Load libptmi.so (built in-tree by ``__graft_entry__.build()`` / ``make lib``).  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise PtmiError(-7, f"{_LIB_PATH} is missing: build it with `make lib` (there is no fallback path)")
    lib = C.CDLL(_LIB_PATH)
    vp, u32 = C.c_void_p, C.c_uint32
    lib.ptmi_setup_context.argtypes = [C.POINTER(vp), C.POINTER(Config)]
    lib.ptmi_initialize_memory.argtypes = [vp, C.POINTER(SceneDesc)]
    lib.ptmi_render.argtypes = [vp, u32, u32]
    lib.ptmi_synchronize.argtypes = [vp]
    lib.ptmi_read_image.argtypes = [vp, vp, vp]
    lib.ptmi_read_display.argtypes = [vp, vp, u32]
    lib.ptmi_snapshot.argtypes = [vp, u32]
    lib.ptmi_render_snapshots.argtypes = [vp, u32, u32, u32]
    lib.ptmi_pin_host_buffer.argtypes = [vp, vp, C.c_size_t]
    lib.ptmi_unpin_host_buffer.argtypes = [vp, vp]
    lib.ptmi_read_snapshot.argtypes = [vp, u32, vp, vp]
    lib.ptmi_write_variance.argtypes = [vp, vp]
    lib.ptmi_write_image.argtypes = [vp, vp, vp]
    lib.ptmi_read_statistics.argtypes = [vp, vp, vp, vp]
    lib.ptmi_clear.argtypes = [vp]
    lib.ptmi_release.argtypes = [vp]
    lib.ptmi_release.restype = None
    lib.ptmi_get_counters.argtypes = [vp, C.POINTER(Counters)]
    lib.ptmi_get_scheduler_stats.argtypes = [vp, C.POINTER(SchedulerStats)]
    lib.ptmi_get_invariant_checks.argtypes = [vp, C.POINTER(InvariantChecks)]
    lib.ptmi_validate_scene.argtypes = [C.POINTER(Config), C.POINTER(SceneDesc)]
    lib.ptmi_literal_kernel_reason.argtypes = [vp]
    lib.ptmi_literal_kernel_reason.restype = C.c_char_p
    lib.ptmi_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(u32)]
    lib.ptmi_reduce_path.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.ptmi_set_stream.argtypes = [vp, vp]
    lib.ptmi_device_accumulators.argtypes = [vp, C.POINTER(vp), C.POINTER(vp)]
    lib.ptmi_bind_accumulators.argtypes = [vp, vp, vp]
    lib.ptmi_read_variance.argtypes = [vp, vp]
    lib.ptmi_device_variance.argtypes = [vp, C.POINTER(vp)]
    lib.ptmi_last_error.argtypes = [vp]
    lib.ptmi_last_error.restype = C.c_char_p
    lib.ptmi_bvh_create.argtypes = [vp, u32, vp, C.POINTER(u32), C.POINTER(u32)]
    lib.ptmi_device_share.argtypes = [u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    lib.ptmi_device_share.restype = None
    _lib = lib
    return lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None and a.size else None


def _f4(v):
    v = np.asarray(v, np.float32)
    return Float4(float(v[0]), float(v[1]), float(v[2]), float(v[3]))


def device_share(first_iteration, n_iterations, k, n_devices):
    """(first_k, n_k): the iteration ids device ``k`` of ``n_devices`` takes from a render call (ids = k mod n_devices)."""
    lib = load_library()
    f, n = C.c_uint32(0), C.c_uint32(0)
    lib.ptmi_device_share(first_iteration, n_iterations, k, n_devices, C.byref(f), C.byref(n))
    return f.value, n.value


def bvh_create(scene):
    """``BVH_Create(globalVars)`` (PathTracer_BVH.cpp:12-37): builds ``scene.bvh`` and reorders
    ``scene.triangulation`` in place, exactly like the reference (host code, no GPU needed)."""
    lib = load_library()
    tris = np.ascontiguousarray(scene.triangulation)
    if tris.dtype != S.Triangle:
        raise PtmiError(-1, "triangulation must have the 336-byte structs.Triangle dtype")
    n = len(tris)
    nodes = np.zeros(max(2 * n - 1, 1), dtype=S.Node)
    size, depth = C.c_uint32(0), C.c_uint32(0)
    rc = lib.ptmi_bvh_create(_ptr(tris), n, _ptr(nodes), C.byref(size), C.byref(depth))
    if rc:
        raise PtmiError(rc, lib.ptmi_last_error(None).decode())
    scene.triangulation = tris
    # byte-level copy (numpy's .copy() of a padded struct dtype leaves the padding bytes undefined)
    scene.bvh = np.frombuffer(bytearray(nodes[:size.value].tobytes()), dtype=S.Node)
    scene.bvhMaxDepth = depth.value
    return scene


def scene_desc(scene):
    """(ptmi_scene struct pointing into contiguous copies of the scene's arrays, those arrays - keep them alive while it is used)"""
    arrays = dict(bvh=np.ascontiguousarray(scene.bvh), tri=np.ascontiguousarray(scene.triangulation),
                  lights=np.ascontiguousarray(scene.lights), mats=np.ascontiguousarray(scene.materiaux),
                  tex=np.ascontiguousarray(scene.textures), texels=np.ascontiguousarray(scene.texturesData),
                  sky=np.ascontiguousarray(scene.sky))
    assert arrays["bvh"].dtype == S.Node and arrays["tri"].dtype == S.Triangle
    d = SceneDesc()
    d.struct_size = C.sizeof(SceneDesc)
    d.bvh, d.bvh_size = _ptr(arrays["bvh"]), len(arrays["bvh"])
    d.triangulation, d.triangulation_size = _ptr(arrays["tri"]), len(arrays["tri"])
    d.lights, d.lights_size = _ptr(arrays["lights"]), len(arrays["lights"])
    d.materiaux, d.materiaux_size = _ptr(arrays["mats"]), len(arrays["mats"])
    d.textures, d.textures_size = _ptr(arrays["tex"]), len(arrays["tex"])
    d.textures_data, d.textures_data_size = _ptr(arrays["texels"]), len(arrays["texels"])
    d.sky = arrays["sky"].ctypes.data_as(C.c_void_p)
    d.camera_position, d.camera_direction = _f4(scene.cameraPosition), _f4(scene.cameraDirection)
    d.camera_right, d.camera_up = _f4(scene.cameraRight), _f4(scene.cameraUp)
    return d, arrays


def validate_scene(scene, image_width, image_height, ray_max_depth, sampler=S.JITTERED, super_sampling=False, flags=0):
    """ptmi_validate_scene: would initialize_memory accept this scene on such a context?  Host-only (no GPU needed).  Raises
    PtmiError with the library's message if not."""
    lib = load_library()
    cfg = Config(C.sizeof(Config), 0, image_width, image_height, ray_max_depth, scene.lightsSize, sampler, 1 if super_sampling else 0, flags)
    d, keep = scene_desc(scene)
    rc = lib.ptmi_validate_scene(C.byref(cfg), C.byref(d))
    del keep
    if rc:
        raise PtmiError(rc, lib.ptmi_last_error(None).decode())


class Backend:
    """One render context = the file-scope OpenCL objects of PathTracer_OpenCL.cpp:19-47."""

    def __init__(self):
        self._lib = load_library()
        self._ctx = C.c_void_p(None)
        self.cfg = None
        self._scene_keepalive = None

    # -- error convention ---------------------------------------------------------------
    def _check(self, rc):
        if rc:
            raise PtmiError(rc, self._lib.ptmi_last_error(self._ctx).decode())

    # -- OpenCL_SetupContext(globalVars, sampler), OpenCL.cpp:316-402 ---------------------
    def setup_context(self, image_width, image_height, ray_max_depth, lights_size, sampler=S.JITTERED,
                      super_sampling=False, device=0, flags=0, devices=None):
        """``devices``: list of HIP ordinals that share every render (spp shards inside the library, summed on the
        first one); None = the single ``device``, like the reference's devices[0] (OpenCL.cpp:363-366)."""
        if self._ctx:
            self.release()
        cfg = Config(C.sizeof(Config), device, image_width, image_height, ray_max_depth, lights_size, sampler,
                     1 if super_sampling else 0, flags)
        if devices:
            cfg.device = devices[0]
            cfg.n_devices = len(devices)
            for i, o in enumerate(devices):
                cfg.devices[i] = o
        ctx = C.c_void_p(None)
        rc = self._lib.ptmi_setup_context(C.byref(ctx), C.byref(cfg))
        if rc:
            raise PtmiError(rc, self._lib.ptmi_last_error(None).decode())
        self._ctx, self.cfg = ctx, cfg
        return self

    # -- OpenCL_InitializeMemory(globalVars), OpenCL.cpp:149-198 --------------------------
    def initialize_memory(self, scene):
        d, keep = scene_desc(scene)
        self._check(self._lib.ptmi_initialize_memory(self._ctx, C.byref(d)))
        del keep
        return self

    # -- one launch of the loop body of OpenCL_RunKernel, generalised to a range ----------
    def render(self, first_iteration, n_iterations):
        self._check(self._lib.ptmi_render(self._ctx, first_iteration, n_iterations))

    def synchronize(self):
        self._check(self._lib.ptmi_synchronize(self._ctx))

    def clear(self):
        self._check(self._lib.ptmi_clear(self._ctx))

    def read_image(self, out=None):
        """(imageColor float32[H,W,4], imageRayNb float32[H,W]) -- the per-image readback, OpenCL.cpp:97-98."""
        h, w = self.cfg.image_height, self.cfg.image_width
        color, count = out if out is not None else (np.empty((h, w, 4), np.float32), np.empty((h, w), np.float32))
        self._check(self._lib.ptmi_read_image(self._ctx, _ptr(color), _ptr(count)))
        return color, count

    def pin_host_buffer(self, array):
        """Page-lock a numpy array that read_image / read_snapshot will fill repeatedly (keep it alive until unpin / release)."""
        self._check(self._lib.ptmi_pin_host_buffer(self._ctx, _ptr(array), array.nbytes))

    def unpin_host_buffer(self, array):
        self._check(self._lib.ptmi_unpin_host_buffer(self._ctx, _ptr(array)))

    def snapshot(self, slot=0):
        """Queue a device-side copy of the accumulators behind the launches issued so far (ptmi_snapshot)."""
        self._check(self._lib.ptmi_snapshot(self._ctx, slot))

    def render_snapshots(self, first_iteration, n_iterations, first_slot=0):
        """render() that also leaves a snapshot after each of its iterations (slots first_slot, first_slot + 1, ...)."""
        self._check(self._lib.ptmi_render_snapshots(self._ctx, first_iteration, n_iterations, first_slot))

    def read_snapshot(self, slot=0, out=None):
        """Wait for snapshot ``slot`` only and return it; ``out`` = (color, count) arrays to fill (DMA'd into directly
        if they were page-locked with pin_host_buffer)."""
        h, w = self.cfg.image_height, self.cfg.image_width
        color, count = out if out is not None else (np.empty((h, w, 4), np.float32), np.empty((h, w), np.float32))
        self._check(self._lib.ptmi_read_snapshot(self._ctx, slot, _ptr(color), _ptr(count)))
        return color, count

    def write_image(self, image_color=None, image_ray_nb=None):
        """Load the accumulators (resume a saved render); the inverse of read_image."""
        c = None if image_color is None else np.ascontiguousarray(image_color, np.float32)
        n = None if image_ray_nb is None else np.ascontiguousarray(image_ray_nb, np.float32)
        self._check(self._lib.ptmi_write_image(self._ctx, None if c is None else _ptr(c), None if n is None else _ptr(n)))

    def read_display(self):
        """Padded B,G,R scanlines uint8[H, (3W+3)&~3] quantised on the device as the reference's viewer does on the
        host (ConvertRGBAToBMPBuffer, Alone/PathTracer_bitmap.cpp:237-286)."""
        h, w = self.cfg.image_height, self.cfg.image_width
        stride = (3 * w + 3) & ~3
        out = np.empty((h, stride), np.uint8)
        self._check(self._lib.ptmi_read_display(self._ctx, _ptr(out), stride))
        return out

    def read_statistics(self):
        """(rayDepths[D+1], rayIntersectedBBx[5000], rayIntersectedTri[5000]), OpenCL.cpp:110-112."""
        d = np.zeros(self.cfg.ray_max_depth + 1, np.uint32)
        b = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
        t = np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)
        self._check(self._lib.ptmi_read_statistics(self._ctx, _ptr(d), _ptr(b), _ptr(t)))
        return d, b, t

    def counters(self):
        c = Counters()
        self._check(self._lib.ptmi_get_counters(self._ctx, C.byref(c)))
        return c.as_dict()

    def scheduler_stats(self):
        s = SchedulerStats()
        self._check(self._lib.ptmi_get_scheduler_stats(self._ctx, C.byref(s)))
        return s.as_dict()

    def invariant_checks(self):
        """Failures of the reference's -D LOG_INFO device-side checks, counted by FLAG_SCHEDULER_STATS builds (ptmi.h)."""
        s = InvariantChecks()
        self._check(self._lib.ptmi_get_invariant_checks(self._ctx, C.byref(s)))
        return s.as_dict()

    def literal_kernel_reason(self):
        """None, or why the uploaded scene is rendered by the one-path-per-lane kernel (records on which the reference's
        triangle test yields NaN distances: ptmi.h)."""
        r = self._lib.ptmi_literal_kernel_reason(self._ctx)
        return r.decode() if r else None

    def reduce_path(self):
        """How a multi-device context sums its partial images: dict(rccl_state, communicators, nccl_version) (ptmi_reduce_path)"""
        a, b, c = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self._lib.ptmi_reduce_path(self._ctx, C.byref(a), C.byref(b), C.byref(c)))
        return {"rccl_state": a.value, "communicators": b.value, "nccl_version": c.value}

    def kernel_time(self):
        """(total_ms, launches) of the integrator kernel since the last call (HIP events on its stream)."""
        ms, n = C.c_double(0), C.c_uint32(0)
        self._check(self._lib.ptmi_kernel_time(self._ctx, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_stream(self, hip_stream):
        self._check(self._lib.ptmi_set_stream(self._ctx, C.c_void_p(hip_stream)))

    def device_accumulators(self):
        a, b = C.c_void_p(None), C.c_void_p(None)
        self._check(self._lib.ptmi_device_accumulators(self._ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def read_variance(self):
        """imageV float32[H,W,4] (SUPER_SAMPLING only): per-channel sum of squared deviations, FullKernel.cl:1346-1349."""
        h, w = self.cfg.image_height, self.cfg.image_width
        v = np.empty((h, w, 4), np.float32)
        self._check(self._lib.ptmi_read_variance(self._ctx, _ptr(v)))
        return v

    def write_variance(self, image_v):
        v = np.ascontiguousarray(image_v, np.float32)
        self._check(self._lib.ptmi_write_variance(self._ctx, _ptr(v)))

    def device_variance(self):
        a = C.c_void_p(None)
        self._check(self._lib.ptmi_device_variance(self._ctx, C.byref(a)))
        return a.value

    def bind_accumulators(self, d_color, d_count):
        self._check(self._lib.ptmi_bind_accumulators(self._ctx, C.c_void_p(d_color), C.c_void_p(d_count)))

    # -- OpenCL_RunKernel(globalVars, UpdateWindowFunc, numImagesToRender, t1, t2, t3), OpenCL.cpp:56-140
    def run_kernel(self, update_window_func=None, num_images_to_render=1, images_per_launch=1):
        """The reference's render loop: launch, read the accumulators back, call the display callback,
        once per image; statistics after the loop; then release.  Returns
        (imageColor, imageRayNb, (rayDepths, bbx, tri), (pathTracingTime, memoryTime, displayTime)) with the
        times in seconds.  ``images_per_launch`` > 1 batches iterations per launch (callback every batch)."""
        t_path = t_mem = t_disp = 0.0
        color = count = None
        image_id = 0
        while image_id < num_images_to_render:
            n = min(images_per_launch, num_images_to_render - image_id)
            t0 = time.perf_counter()
            self.render(image_id, n)
            self.synchronize()
            t1 = time.perf_counter()
            color, count = self.read_image()
            t2 = time.perf_counter()
            if update_window_func is not None:
                update_window_func()  # return value ignored, as in OpenCL.cpp:103
            t3 = time.perf_counter()
            t_path += t1 - t0
            t_mem += t2 - t1
            t_disp += t3 - t2
            image_id += n
        stats = self.read_statistics()
        self.release()
        return color, count, stats, (t_path, t_mem, t_disp)

    def release(self):
        if self._ctx:
            self._lib.ptmi_release(self._ctx)
            self._ctx = C.c_void_p(None)

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def render_scene(scene, width, height, ray_max_depth, n_iterations, first_iteration=0, sampler=S.JITTERED, device=0,
                 flags=0, super_sampling=False, devices=None):
    """Convenience used by tests/bench: full life cycle for one iteration range."""
    be = Backend().setup_context(width, height, ray_max_depth, scene.lightsSize, sampler, super_sampling=super_sampling,
                                 device=device, flags=flags, devices=devices)
    try:
        be.initialize_memory(scene)
        be.render(first_iteration, n_iterations)
        color, count = be.read_image()
        stats = be.read_statistics()
        counters = be.counters()
    finally:
        be.release()
    return color, count, stats, counters
