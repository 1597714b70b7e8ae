"""The integrator in the arithmetic of the build the reference's OWN build line produces (GPU).

`OpenCL_BuildOptions` (Controleur/PathTracer_OpenCL.cpp:292-314) passes no floating-point option, so the reference runs its
OpenCL compiler's DEFAULT arithmetic: `a * b + c` written in one expression is one fused multiply-add, a division goes
through v_rcp_f32 of the divisor's mantissa, a square root is v_sqrt_f32.  `PTMI_FLAG_DEFAULT_ARITHMETIC` selects a build
of the integrator's device code that restates exactly that (csrc/ptmi_device.hpp), and the oracle has the same second
build (oracle/build/libpt_oracle_da.so).  Here:

  * operator level: the product's fdiv / frcp / constant-divisor fdiv / fsqrt / length / mad == what the image's OpenCL
    compiler emits for `/`, `sqrt`, `length`, `a*b+c` (oracle/arith_probe.cl, our own source compiled with the reference's
    flags) == the oracle's table-driven host emulation, on millions of operands;
  * kernel level: images, sample counts and the three histograms == the reference kernel's default build
    (oracle/_ref/ref_kernel_*.hsaco: unmodified source, the reference's own options), bit for bit, on the seven parity
    cases, the twelve one-feature scenes and at BASELINE's full sizes;
  * oracle level: HIP (both kernels) == the oracle's default-arithmetic build, bit for bit.
"""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_ffi as O
from opencl_pathtracer_amd import render_scene, scenes, bvh_create, backend, structs as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "oracle", "build", "libdevice_math_probe.so")
PROBE_CL = os.path.join(ROOT, "oracle", "build", "arith_probe.hsaco")
DA = backend.FLAG_DEFAULT_ARITHMETIC
OPS = ["a / b", "1.0f / a", "a / 255.f", "a / 3.f", "a / 1.55f", "a / 1920", "a / 90", "sqrt(a)", "length(a, b, a/2, 0)", "a * b + 1"]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p)


def _same_bits(x, y):
    """bit equality, any NaN equal to any NaN"""
    return (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))


def test_default_arithmetic_operators():
    if not (os.path.exists(PROBE) and os.path.exists(PROBE_CL)):
        pytest.fail("oracle/build/libdevice_math_probe.so / arith_probe.hsaco not built (make -C oracle probe)")
    rs = np.random.RandomState(5)
    n_rand = 1 << 20
    def wide(n):
        return (rs.choice([-1.0, 1.0], n) * rs.uniform(1, 2, n) * np.exp2(rs.randint(-149, 128, n).astype(np.float64))).astype(np.float32)
    special = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, -1.0, 2.0, 0.5, 1e-45, -1e-45, 1.1754942e-38, 1.1754944e-38,
                        3.4028235e38, 255.0, 3.0, 1.55, 1e-39, 5e-42], np.float32)
    sa, sb = np.meshgrid(special, special)
    a = np.concatenate([wide(n_rand), rs.uniform(-2, 2, n_rand).astype(np.float32), rs.uniform(0, 1, n_rand).astype(np.float32), sa.ravel()])
    b = np.concatenate([wide(n_rand), rs.uniform(-2, 2, n_rand).astype(np.float32), rs.uniform(0.5, 4, n_rand).astype(np.float32), sb.ravel()])
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    n = len(a)
    product, compiler, hw = np.empty((10, n), np.float32), np.empty((10, n), np.float32), np.empty((2, n), np.float32)
    lib = C.CDLL(PROBE)
    lib.device_arith_probe.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
    assert lib.device_arith_probe(PROBE_CL.encode(), _vp(a), _vp(b), n, _vp(product), _vp(compiler), _vp(hw)) == 0
    for k, op in enumerate(OPS):
        bad = np.flatnonzero(~_same_bits(product[k], compiler[k]))
        assert len(bad) == 0, (f"{op}: the product's default-arithmetic operator differs from the OpenCL compiler's on {len(bad)} of {n} "
                               f"operands, e.g. a={a[bad[0]]!r} b={b[bad[0]]!r}: {product[k][bad[0]]!r} vs {compiler[k][bad[0]]!r}")
    # ... and the oracle's host emulation (tables of v_rcp_f32 / v_sqrt_f32 measured on this GPU) gives the same bits
    ol = O.oracle(default_arithmetic=True)
    host = np.empty((10, n), np.float32)
    ol.pto_arith_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    ol.pto_arith_probe(_vp(a), _vp(b), n, _vp(host))
    for k, op in enumerate(OPS):
        bad = np.flatnonzero(~_same_bits(host[k], compiler[k]))
        assert len(bad) == 0, (f"{op}: the oracle's default-arithmetic build differs from the OpenCL compiler's on {len(bad)} of {n} "
                               f"operands, e.g. a={a[bad[0]]!r} b={b[bad[0]]!r}: {host[k][bad[0]]!r} vs {compiler[k][bad[0]]!r}")
    # the strict build of the oracle is something else: the two arithmetics are really different
    strict = np.empty((10, n), np.float32)
    os_ = O.oracle(default_arithmetic=False)
    os_.pto_arith_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    os_.pto_arith_probe(_vp(a), _vp(b), n, _vp(strict))
    assert (~_same_bits(strict[0], host[0])).mean() > 0.01 and (~_same_bits(strict[7], host[7])).mean() > 0.01


def _assert_equal_to_reference(ours, ref, what):
    color, count, (dep, bbx, tri), _ = ours
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = ref
    assert np.array_equal(count, r_count), f"{what}: sample counts differ"
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri), f"{what}: histograms differ"
    bad = np.argwhere(color.view(np.uint32) != r_color.view(np.uint32))
    assert len(bad) == 0, f"{what}: {len(bad)} channel values differ from the reference's default build, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("case", list(cases.CASES))
def test_bit_exact_vs_reference_default_build(case, scene_factory):
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present (built only where the reference tree exists)")
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    spp = 16
    ref = O.ref_gpu_render(case, sc, w, h, d, spp)
    _assert_equal_to_reference(render_scene(sc, w, h, d, spp, sampler=sampler, flags=DA), ref, case)
    if case in ("cornell_64x48_d4", "matmix_96x96_d8", "tris20k_96x64_d6"):  # the one-path-per-lane kernel too
        _assert_equal_to_reference(render_scene(sc, w, h, d, spp, sampler=sampler, flags=DA | backend.FLAG_MEGAKERNEL), ref, case + " (megakernel)")


@pytest.mark.parametrize("feature", scenes.FEATURES)
def test_every_feature_bit_exact_vs_reference_default_build(feature):
    """One kernel feature per scene (materials, textures, light types, sky, two-sided / fallback normals), 256 spp = 1M paths
    each: all five material branches, every light type, every division / square root / contraction site of the kernel."""
    case, w, h, d = "feat_64x64_d8", 64, 64, 8
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    sc = bvh_create(scenes.build("feat_" + feature, w, h))
    ref = O.ref_gpu_render(case, sc, w, h, d, 256)
    _assert_equal_to_reference(render_scene(sc, w, h, d, 256, flags=DA), ref, feature)


FULL_SIZE = {"cornell_512x512_d4": ("cornell", S.JITTERED, 512, 512, 4, 8),          # BASELINE configs[0]
             "matmix_3840x2160_d16": ("matmix", S.JITTERED, 3840, 2160, 16, 1),     # the toy material mix at configs[4]'s size: 4K, 16 bounces, 3 lights
             # the stand-in for BASELINE configs[4] (SURVEY 8d Config 5): 1.09 M textured triangles, 4 x 1024^2 textures, 6 x 512^2 sky
             "mayalike_3840x2160_d16": ("mayalike", S.JITTERED, 3840, 2160, 16, 1),
             "tris1m_1920x1080_d10": ("tris1m", S.JITTERED, 1920, 1080, 10, 2),   # BASELINE configs[2]: the bench workload
             "cornell_1920x1080_d8": ("cornell", S.JITTERED, 1920, 1080, 8, 4)}     # BASELINE configs[1]


@pytest.mark.parametrize("case", list(FULL_SIZE))
def test_full_size_bit_exact_vs_reference_default_build(case):
    """BASELINE's own image size, scene and ray depth: all 2 M pixels, the sample counts and the three histograms equal the
    reference kernel as its own build line compiles it, run beside the integrator on this GPU."""
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present (built only where the reference tree exists)")
    name, sampler, w, h, d, spp = FULL_SIZE[case]
    sc = bvh_create(scenes.build(name, w, h))
    ref = O.ref_gpu_render(case, sc, w, h, d, spp)
    _assert_equal_to_reference(render_scene(sc, w, h, d, spp, sampler=sampler, flags=DA), ref, case)


@pytest.mark.parametrize("name,w,h,d,lanes,waves_per_cu", [("tris1m", 1920, 1080, 10, 256, 20), ("mayalike", 3840, 2160, 16, 64, 17)])
def test_persistent_grid_of_the_production_kernels(name, w, h, d, lanes, waves_per_cu):
    """The headline scene's tree is 22 levels deep: five workgroups of 256 lanes fill a CU's LDS to the last allocation granule
     - 192 bytes more of static LDS cost the fifth workgroup and 9 % in round 4, without a test noticing.  Deeper
    trees (the configs[4] stand-in: 23 levels) run workgroups of 64 lanes, of which a CU holds 17 at least."""
    import os
    import torch
    if lanes == 64 and os.environ.get("PTMI_GENERIC_TRIANGLES"):
        pytest.skip("workgroups of 64 lanes exist for the two production instantiations only (precomputed triangle records)")
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    sc = bvh_create(scenes.build(name, w, h))
    be = backend.Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, flags=DA)
    try:
        be.initialize_memory(sc)
        be.render(0, 1)
        be.synchronize()
        grid = be.scheduler_stats()
    finally:
        be.release()
    assert grid["workgroup_lanes"] == lanes, grid
    assert grid["resident_workgroups"] * lanes // 64 >= waves_per_cu * cus, (grid, cus)


def test_the_one_bounce_the_reference_leaves_undefined():
    """The reference's water material refracts when random() >= its Fresnel fraction (cl:836-843).  On total internal reflection
    that fraction is 1 and Material_FresnelWaterReflectionFraction has returned (cl:237) BEFORE it writes the refraction direction
    and factor (cl:249-251) - and random() returns exactly 1.0 for the 64 seeds nearest 2^31 (header.cl:246-253: the float
    conversion rounds them up).  So one interior water hit in ~10^8 refracts along an UNINITIALISED direction with an uninitialised
    factor: undefined in the reference's source; its compiled kernel reads stale registers.  The configs[4] stand-in
    at full size holds one such path in its first 64 iterations - pixel (3075, 393), iteration 25 - found by round 4's full-size
    comparison: that pixel is the ONLY difference from the reference kernel in that iteration, the statistics build counts exactly
    that one bounce, and the oracle agrees with the integrator about it."""
    case, w, h, d, it = "mayalike_3840x2160_d16", 3840, 2160, 16, 25
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    sc = bvh_create(scenes.build("mayalike", w, h))
    ref = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it)
    ours = render_scene(sc, w, h, d, 1, first_iteration=it, flags=DA)
    bad = np.argwhere((ours[0].view(np.uint32) != ref[0].view(np.uint32)).any(-1))
    assert bad.tolist() == [[393, 3075]], bad[:8].tolist()
    assert np.array_equal(ours[1], ref[1])
    be = backend.Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, flags=DA | backend.FLAG_SCHEDULER_STATS)
    try:
        be.initialize_memory(sc)
        be.render(it, 1)
        be.synchronize()
        checks = be.invariant_checks()
        stats_image, _ = be.read_image()
    finally:
        be.release()
    assert checks["refraction_undefined_in_reference"] == 1, checks
    assert np.array_equal(stats_image.view(np.uint32), ours[0].view(np.uint32))
    _, o_radiance = O.oracle_trace(sc, w, h, d, 3075, 393, it, default_arithmetic=True)
    assert np.array_equal(np.asarray(o_radiance, np.float32).view(np.uint32), ours[0][393, 3075].view(np.uint32))


def test_four_million_triangles_bit_exact_vs_reference_default_build():
    """A tree of depth 24 (4M random triangles, 427 MB of records: four workgroups per CU instead of five, leaves and nodes far
    beyond the L2s) through the same comparison: the reference's code object is specialised on sampler, image size, ray depth and
    light count only, so the 1M-triangle configuration's kernel renders this scene too.  1920x1080, 1 spp."""
    case = "tris1m_1920x1080_d10"
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present (built only where the reference tree exists)")
    w, h, d = 1920, 1080, 10
    sc = bvh_create(scenes.build("tris4m", w, h))
    assert sc.bvhMaxDepth >= 23
    ref = O.ref_gpu_render(case, sc, w, h, d, 1)
    _assert_equal_to_reference(render_scene(sc, w, h, d, 1, flags=DA), ref, "tris4m")


@pytest.mark.parametrize("case", ["cornell_64x48_d4", "cornell_64x48_d4_uni", "matmix_96x96_d8", "tris20k_96x64_d6"])
def test_default_arithmetic_hip_equals_oracle(case, scene_factory):
    """HIP (wavefront and one-path-per-lane kernels) == the oracle's default-arithmetic build, bit for bit - the oracle's
    second build is pinned to the reference's default build through the integrator and directly by
    tests/test_oracle_golden.py (committed outputs of that build)."""
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    spp = 8
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, spp, sampler=sampler, default_arithmetic=True)
    for flags in (DA, DA | backend.FLAG_MEGAKERNEL):
        color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, spp, sampler=sampler, flags=flags)
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)), f"flags {flags}"
        assert np.array_equal(count, o_count) and np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
        assert counters == totals
    # and the two arithmetics are really different renders
    s_color, _, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler)
    assert not np.array_equal(s_color.view(np.uint32), o_color.view(np.uint32))


def test_random_sampler_vs_reference_default_build(scene_factory):
    """SAMPLE_RANDOM: a sample lands on the pixel its two random numbers name (cl:1137-1141, :1333-1336) and the reference adds it
    there with a plain read-modify-write - work-items that hit one pixel at the same time lose updates.  The integrator adds
    atomically.  So: its sample counts equal the serial oracle's exactly (every sample on the pixel the reference's arithmetic
    names), the reference's own counts are those minus what its race lost - never more, and nearly all of them - and the images
    agree where no update was lost."""
    case = "cornell_64x48_d4_rnd"
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    w, h, d, n = 64, 48, 4, 8
    sc = scene_factory("cornell", w, h)
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, n)
    color, count, (dep, bbx, tri), _ = render_scene(sc, w, h, d, n, sampler=S.RANDOM, flags=DA)
    o_color, o_count, _, _ = O.oracle_render(sc, w, h, d, n, sampler=S.RANDOM, default_arithmetic=True)
    assert np.array_equal(count, o_count) and np.allclose(color, o_color, rtol=1e-5, atol=1e-5)
    # the paths themselves are the reference's, exactly: histograms are atomic in the reference too
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    assert (r_count <= count).all() and r_count.sum() >= 0.97 * count.sum()
    same = r_count == count  # pixels where the reference lost no count (measured: 89 % of them after 8 iterations)
    assert same.mean() > 0.8
    # the sums: the colour and the count of a pixel are separate read-modify-writes in the reference, so it can lose either;
    # radiance is non-negative, so what it keeps never exceeds the atomic sum, and most pixels keep everything
    assert (r_color[..., :3] <= color[..., :3] * (1 + 1e-5) + 1e-5).all()
    close = np.isclose(color, r_color, rtol=1e-5, atol=1e-5).all(-1)
    assert close[same].mean() > 0.75 and close.mean() > 0.6  # (measured: 0.85 and 0.76; the race decides)


def test_random_sampler_paths_with_nan_rays_are_traced_again(monkeypatch):
    """fuzz47r: a few paths in a million scatter into a direction that is not a number (a refraction's square root of a negative,
    cl:235,249); the reference then accepts triangles with NaN distances and keeps the LAST that passes.  With the RANDOM sampler
    nothing is staged, so the wavefront kernel LISTS such a path in the launch's job-counter block and redo_random_kernel traces it
    again with the literal loops and adds it atomically (round 4; before, the path kept the ordered minimum).  The paths of a
    RANDOM-sampler render are a function of (work-item, iteration) like any other, and the reference's histograms are atomics:
    all three equal the reference kernel's, built -D SAMPLE_RANDOM with its own options - and so do the one-path-per-lane
    kernel's and the oracle's totals."""
    import warnings
    case, w, h, d, spp = "fuzz_96x64_d10_rnd", 96, 64, 10, 2048
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = bvh_create(scenes.build("fuzz47r_l1", w, h))
    be = backend.Backend().setup_context(w, h, d, 1, S.RANDOM, flags=DA)
    try:
        be.initialize_memory(sc)
        assert be.literal_kernel_reason() is None
        be.render(0, spp)
        color, count = be.read_image()
        dep, bbx, tri = be.read_statistics()
        counters = be.counters()
        retraced = be.scheduler_stats()["paths_retraced"]
    finally:
        be.release()
    assert 0 < retraced < 1000, retraced  # (measured: 37 of 12.6 M)
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, spp)
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    assert (r_count <= count).all() and count.sum() == w * h * spp
    # the one-path-per-lane kernel (the reference's loops as they are written): the same histograms, counts and totals
    lit = render_scene(sc, w, h, d, spp, sampler=S.RANDOM, flags=DA | backend.FLAG_MEGAKERNEL)
    assert all(np.array_equal(a, b) for a, b in zip((dep, bbx, tri), lit[2])) and np.array_equal(count, lit[1]) and counters == lit[3]
    # ... and with the re-trace switched off by an empty list (a launch of the old kind), the histograms differ: the test tests
    monkeypatch.setenv("PTMI_RANDOM_GIVE_UP", "0")
    old = render_scene(sc, w, h, d, spp, sampler=S.RANDOM, flags=DA)
    assert not (np.array_equal(old[2][1], r_bbx) and np.array_equal(old[2][2], r_tri))


FUZZ_SPECIALISATIONS = ((1, "feat_64x64_d8", 64, 64, 8), (3, "matmix_96x96_d8", 96, 96, 8))  # lights, reference code object, W, H, depth


@pytest.mark.parametrize("hostile", [False, True], ids=["importer", "hostile"])
@pytest.mark.parametrize("seed", range(10))
def test_fuzzed_scenes_bit_exact_vs_both_reference_builds(seed, hostile):
    """Scenes drawn from a seed (scenes.fuzz_scene: soup at mixed scales, fans, slivers, coincident and coplanar stacks, flat
    boxes, tilted / zero normals, all material and light types at random, textures down to 1x1 with uv far outside [0, 1],
    random cube maps, a skewed film frame; `hostile`: zero-area triangles, a light on a vertex, coordinates of 1e6): image
    bits (NaNs included), sample counts and the three histograms equal the reference kernel's - its own build with
    PTMI_FLAG_DEFAULT_ARITHMETIC, its strict build without - and the CPU oracle's in both arithmetics."""
    import warnings
    problems = []

    def check(ours, ref, what):
        try:
            _assert_equal_to_reference(ours, ref, what)
        except AssertionError as e:
            problems.append(str(e).split("\n")[0])

    for n_lights, case, w, h, d in FUZZ_SPECIALISATIONS:
        if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
            O.missing_reference("oracle/_ref code objects not present")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")  # (the hostile scenes divide 0 by 0 on purpose, as the importer would)
            sc = bvh_create(scenes.build(f"fuzz{seed}{'h' if hostile else ''}_l{n_lights}", w, h))
        spp = 24
        for strict in (False, True):
            flags = 0 if strict else DA
            what = f"{sc.name} ({'strict' if strict else 'default'} build)"
            ours = render_scene(sc, w, h, d, spp, flags=flags)
            check(ours, O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict), what)
            check(render_scene(sc, w, h, d, spp, flags=flags | backend.FLAG_MEGAKERNEL), ours, what + ": one-path-per-lane kernel vs wavefront kernel")
            o = O.oracle_render(sc, w, h, d, 3, default_arithmetic=not strict)
            check(render_scene(sc, w, h, d, 3, flags=flags), o, what + ": integrator vs CPU oracle")
    assert not problems, "\n".join(problems)


def test_scenes_with_nan_distances_run_the_nan_safe_wavefront_kernel(monkeypatch):
    """A zero-area triangle (N = 0/0 as the importer computes it) is ACCEPTED by the reference for every ray that reaches it, with
    a NaN distance.  Round 4: the context says why the scene is special and renders it with the wavefront kernel's NANSAFE
    instantiation - only the paths that REACH such a record are given up and traced again by the literal loops - for ptmi_render
    and ptmi_render_snapshots alike: the images equal the one-path-per-lane kernel's (the reference's loops as they are written)
    and the reference kernel's own.  With the RANDOM sampler nothing is staged, so the one-path-per-lane kernel takes the scene as
    a whole (and SUPER_SAMPLING, wavefront-only, is refused there)."""
    import warnings
    case, w, h, d = "feat_64x64_d8", 64, 64, 8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        tame, wild = bvh_create(scenes.build("fuzz3_l1", w, h)), bvh_create(scenes.build("fuzz3h_l1", w, h))
    be = backend.Backend().setup_context(w, h, d, 1, S.JITTERED, flags=DA | backend.FLAG_SCHEDULER_STATS)
    try:
        be.initialize_memory(tame)
        assert be.literal_kernel_reason() is None
        be.initialize_memory(wild)
        assert "triangle" in be.literal_kernel_reason()
        be.render_snapshots(0, 6, 0)
        images = [be.read_snapshot(k) for k in range(6)]
        stats = be.read_statistics()
        st = be.scheduler_stats()
        assert st["trips_node"] > 0 and st["paths_retraced"] > 0, "the wavefront kernel ran and handed paths to the literal loops"
        be.initialize_memory(tame)
        assert be.literal_kernel_reason() is None
    finally:
        be.release()
    color, count, ref_stats, _ = render_scene(wild, w, h, d, 6, flags=DA | backend.FLAG_MEGAKERNEL)
    assert np.array_equal(images[5][0].view(np.uint32), color.view(np.uint32)) and np.array_equal(images[5][1], count)
    assert all(np.array_equal(a, b) for a, b in zip(stats, ref_stats))
    c3, n3, _, _ = render_scene(wild, w, h, d, 3, flags=DA | backend.FLAG_MEGAKERNEL)
    assert np.array_equal(images[2][0].view(np.uint32), c3.view(np.uint32)) and np.array_equal(images[2][1], n3)
    # the production instantiation (no statistics), both arithmetics, counters included
    for flags in (DA, 0):
        lit = render_scene(wild, w, h, d, 6, flags=flags | backend.FLAG_MEGAKERNEL)
        ours = render_scene(wild, w, h, d, 6, flags=flags)
        assert np.array_equal(ours[0].view(np.uint32), lit[0].view(np.uint32)) and np.array_equal(ours[1], lit[1]) and ours[3] == lit[3]
        assert all(np.array_equal(a, b) for a, b in zip(ours[2], lit[2]))
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    r_color, r_count, _, _ = O.ref_gpu_render(case, wild, w, h, d, 6)
    assert np.array_equal(r_color.view(np.uint32), color.view(np.uint32)) and np.array_equal(r_count, count)
    # PTMI_LITERAL_KERNEL=1: the scene as a whole on the one-path-per-lane kernel, as before round 4
    monkeypatch.setenv("PTMI_LITERAL_KERNEL", "1")
    forced = render_scene(wild, w, h, d, 6, flags=DA)
    monkeypatch.delenv("PTMI_LITERAL_KERNEL")
    assert np.array_equal(forced[0].view(np.uint32), color.view(np.uint32))
    # RANDOM sampler + SUPER_SAMPLING on such a scene: refused, not rendered differently
    with pytest.raises(backend.PtmiError, match="SUPER_SAMPLING"):
        render_scene(wild, w, h, d, 2, flags=DA, super_sampling=True, sampler=S.RANDOM)


def test_nan_rays_skip_the_walk_over_the_whole_tree(monkeypatch):
    """A path whose FINAL hit is a zero-area triangle shades with NaNs and scatters a ray whose direction is NaN in every
    component; the reference then "hits" every box and accepts every triangle - comparisons only - ten walks over the whole tree.
    The literal loops (the one-path-per-lane kernel, and what re-traces the wavefront kernel's given-up paths) add the walk's
    counts, which the upload has taken once, and make only the walk's LAST triangle test for real: images (NaN bits included),
    counts, histograms and totals equal the walked ones (PTMI_WALK_NAN_RAYS) and the reference kernel's, both builds."""
    case, w, h, d, spp = "tris20k_96x64_d6", 96, 64, 6, 8
    if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
        O.missing_reference("oracle/_ref code objects not present")
    sc = bvh_create(scenes.add_zero_area_triangles(scenes.build("tris20k", w, h), 400))
    for flags, strict in ((DA, False), (0, True)):
        ref = O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict)
        for kernel in (0, backend.FLAG_MEGAKERNEL):
            ours = render_scene(sc, w, h, d, spp, flags=flags | kernel)
            # (the scene tests what it should: whole-tree walks dominate the counts - 20,400 triangles, ~3 queries per path)
            assert ours[3]["triangle_tests"] > 5000 * ours[3]["paths"], ours[3]
            _assert_equal_to_reference(ours, ref, f"closed form, kernel flags {kernel}, strict {strict}")
            monkeypatch.setenv("PTMI_WALK_NAN_RAYS", "1")
            walked = render_scene(sc, w, h, d, spp, flags=flags | kernel)
            monkeypatch.delenv("PTMI_WALK_NAN_RAYS")
            assert ours[3] == walked[3] and np.array_equal(ours[0].view(np.uint32), walked[0].view(np.uint32))
    o = O.oracle_render(sc, w, h, d, 2, default_arithmetic=True)
    ours = render_scene(sc, w, h, d, 2, flags=DA)
    assert np.array_equal(ours[0].view(np.uint32), o[0].view(np.uint32)) and ours[3] == o[3]


def test_super_sampling_on_a_scene_with_nan_distances():
    """SUPER_SAMPLING (wavefront kernel only) on a scene with zero-area triangles - refused before round 4 - equals the reference
    kernel built -D SUPER_SAMPLING with its own options: images, sampling-density map and histograms."""
    import warnings
    case, w, h, d = "cornell_64x48_d4_ss", 64, 48, 4
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wild = bvh_create(scenes.build("fuzz3h_l1", w, h))
    ours = render_scene(wild, w, h, d, 12, flags=DA, super_sampling=True)
    ref = O.ref_gpu_render(case, wild, w, h, d, 12)
    _assert_equal_to_reference(ours, ref, "fuzz3h_l1 with SUPER_SAMPLING")


# (reference code object, lights, sampler, W, H, depth, SUPER_SAMPLING)
FUZZ_OTHER_SPECIALISATIONS = [("cornell_64x48_d4_uni", 1, S.UNIFORM, 64, 48, 4, False), ("matmix_96x96_d8_uni", 3, S.UNIFORM, 96, 96, 8, False),
                              ("tris20k_96x64_d6", 1, S.JITTERED, 96, 64, 6, False), ("cornell_64x48_d4_ss", 1, S.JITTERED, 64, 48, 4, True),
                              ("cornell_128x128_d8", 1, S.JITTERED, 128, 128, 8, False), ("cornell_64x48_d2", 1, S.JITTERED, 64, 48, 2, False),
                              ("tris1m_160x90_d10", 1, S.JITTERED, 160, 90, 10, False), ("cornell_64x48_d1", 1, S.JITTERED, 64, 48, 1, False)]


@pytest.mark.parametrize("seed", range(10, 34))
def test_fuzzed_scenes_other_samplers_sizes_and_depths(seed):
    """More seeds, each through another specialisation of the reference kernel: the UNIFORM sampler, SUPER_SAMPLING (the adaptive
    stop decisions must fall on the same samples), ray depths 1 to 10, image sizes that are no multiple of a tile; odd seeds
    carry the hostile records (not with SUPER_SAMPLING, which such a scene refuses)."""
    import warnings
    case, n_lights, sampler, w, h, d, ss = FUZZ_OTHER_SPECIALISATIONS[seed % len(FUZZ_OTHER_SPECIALISATIONS)]
    hostile = seed % 2 == 1 and not ss
    if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
        O.missing_reference("oracle/_ref code objects not present")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = bvh_create(scenes.build(f"fuzz{seed}{'h' if hostile else ''}_l{n_lights}", w, h))
    spp = 20
    problems = []
    for strict in (False, True):
        flags = 0 if strict else DA
        what = f"{sc.name} through {case} ({'strict' if strict else 'default'} build)"
        ours = render_scene(sc, w, h, d, spp, sampler=sampler, flags=flags, super_sampling=ss)
        for other, label in ((O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict), what),
                             (O.oracle_render(sc, w, h, d, spp, sampler=sampler, super_sampling=ss, default_arithmetic=not strict), what + ": integrator vs CPU oracle")):
            try:
                _assert_equal_to_reference(ours, other, label)
            except AssertionError as e:
                problems.append(str(e).split("\n")[0])
    assert not problems, "\n".join(problems)


@pytest.mark.parametrize("seed", range(40, 56))
def test_fuzzed_scenes_with_records_no_importer_writes(seed):
    """scenes.corrupt_records on top of a fuzzed scene: geometric normals of any length and side, vertices out of order, w
    components that are not 1 (the generic triangle test when they differ inside a triangle), material and light types outside
    their enums, opacities outside [0, 1], negative powers, inverted cones - the raw arrays are the interface, and the
    reference's handling of them is the contract.  Odd seeds add the hostile records."""
    import warnings
    problems = []
    for n_lights, case, w, h, d in FUZZ_SPECIALISATIONS:
        if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
            O.missing_reference("oracle/_ref code objects not present")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sc = bvh_create(scenes.build(f"fuzz{seed}{'h' if seed % 2 else ''}r_l{n_lights}", w, h))
        spp = 16
        for strict in (False, True):
            flags = 0 if strict else DA
            what = f"{sc.name} ({'strict' if strict else 'default'} build)"
            ours = render_scene(sc, w, h, d, spp, flags=flags)
            for other, label in ((O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict), what),
                                 (render_scene(sc, w, h, d, spp, flags=flags | backend.FLAG_MEGAKERNEL), what + ": one-path-per-lane kernel vs wavefront kernel"),
                                 (O.oracle_render(sc, w, h, d, spp, default_arithmetic=not strict), what + ": integrator vs CPU oracle")):
                try:
                    _assert_equal_to_reference(ours, other, label)
                except AssertionError as e:
                    problems.append(str(e).split("\n")[0])
    assert not problems, "\n".join(problems)


@pytest.mark.parametrize("name", ["fuzz3_l1", "fuzz47r_l1", "fuzz3h_l1", "fuzz7h_l1"])
def test_fuzzed_scenes_at_full_size_vs_reference_default_build(name):
    """1920 x 1080, depth 10 (the code object of BASELINE configs[2]: the reference bakes in sizes, depth and light count, not
    the scene): thousands of triangles in clusters with every material type and textures - the GENERAL shading specialisation
    of the wavefront kernel at full size, which the two BASELINE scenes (plain ones) do not reach -, the same with corrupted
    records (where a few paths in a million meet a ray that is not a number), and two hostile ones (zero-area triangles: NaN
    distances) through the wavefront kernel's NANSAFE instantiation, against BOTH builds of the reference.  All 2 M pixels, counts
    and histograms equal."""
    import warnings
    case, w, h, d, spp = "tris1m_1920x1080_d10", 1920, 1080, 10, 2
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = bvh_create(scenes.build(name, w, h))
    ours = render_scene(sc, w, h, d, spp, flags=DA)
    _assert_equal_to_reference(ours, O.ref_gpu_render(case, sc, w, h, d, spp), name)
    if "h" in name.split("_")[0]:
        if not O.have_ref_kernel(case, strict=True):
            O.missing_reference("oracle/_ref strict code object not present")
        _assert_equal_to_reference(render_scene(sc, w, h, d, spp, flags=0), O.ref_gpu_render(case, sc, w, h, d, spp, strict=True), name + " (strict)")
        be = backend.Backend().setup_context(w, h, d, 1, S.JITTERED, flags=DA | backend.FLAG_SCHEDULER_STATS)
        try:
            be.initialize_memory(sc)
            assert be.literal_kernel_reason() is not None
            be.render(0, 1)
            be.synchronize()
            st = be.scheduler_stats()
        finally:
            be.release()
        # the wavefront kernel rendered it, and only the paths that reached a bad record went to the literal loops
        # (fuzz7h's bad records lie outside this camera's view at 1920 x 1080: nothing to trace again there)
        assert st["trips_node"] > 0 and st["paths_retraced"] < w * h // 2 and (st["paths_retraced"] > 0 or name != "fuzz3h_l1"), st
    if name == "fuzz47r_l1":
        # 15 of its 4 M paths scatter into a direction that is not a number (a refraction's square root of a negative): the
        # wavefront kernel gives those up and the literal loops trace them again behind the launch - the reference's pixels
        # (above), and the totals of a render in which nothing was given up
        literal = render_scene(sc, w, h, d, spp, flags=DA | backend.FLAG_MEGAKERNEL)
        assert ours[3] == literal[3] and np.array_equal(ours[0].view(np.uint32), literal[0].view(np.uint32))
        be = backend.Backend().setup_context(w, h, d, 1, S.JITTERED, flags=DA)
        try:
            be.initialize_memory(sc)
            be.render(0, spp)
            be.synchronize()
            retraced = be.scheduler_stats()["paths_retraced"]
        finally:
            be.release()
        assert 0 < retraced < 100, retraced  # (measured: 15 of 4.1 M)


@pytest.mark.parametrize("seed", range(60, 72))
def test_fuzzed_scenes_with_corrupted_trees(seed):
    """scenes.corrupt_tree on the built tree: boxes that do not bound their subtree, inverted, with NaN or infinite faces or marked
    empty, foreign split axes, leaves that lost a triangle (or all).  Structurally valid, so the integrator takes it - and must
    then walk it exactly as the reference does: image bits, counts, histograms, both builds, both kernels, and the oracle."""
    import warnings
    problems = []
    for n_lights, case, w, h, d in FUZZ_SPECIALISATIONS:
        if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
            O.missing_reference("oracle/_ref code objects not present")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sc = scenes.corrupt_tree(bvh_create(scenes.build(f"fuzz{seed}{'h' if seed % 3 == 0 else ''}{'r' if seed % 2 else ''}_l{n_lights}", w, h)), seed)
        spp = 16
        for strict in (False, True):
            flags = 0 if strict else DA
            what = f"{sc.name} ({'strict' if strict else 'default'} build)"
            ours = render_scene(sc, w, h, d, spp, flags=flags)
            for other, label in ((O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict), what),
                                 (render_scene(sc, w, h, d, spp, flags=flags | backend.FLAG_MEGAKERNEL), what + ": one-path-per-lane kernel vs wavefront kernel"),
                                 (O.oracle_render(sc, w, h, d, spp, default_arithmetic=not strict), what + ": integrator vs CPU oracle")):
                try:
                    _assert_equal_to_reference(ours, other, label)
                except AssertionError as e:
                    problems.append(str(e).split("\n")[0])
    assert not problems, "\n".join(problems)
