"""GPU parity: the HIP integrator (through the C ABI) against the CPU oracle and against the reference itself.

Bars:
  * HIP vs oracle, JITTERED / UNIFORM samplers: BIT-EXACT images, histograms and counters (same numerics
    contract, same accumulation order).
  * HIP vs oracle, RANDOM sampler: samples land on arbitrary pixels and are added atomically (the reference
    races there), so only fp32 summation order differs: per-channel RMS <= 1e-6, counts exact.
  * HIP vs the reference kernel run on the same GPU (code object built from the unmodified .cl):
    per-channel RMS of sum/n <= 1e-4 -- the tolerance BASELINE.json's north_star states.
"""
import numpy as np
import pytest

import cases
import oracle_ffi as O
from opencl_pathtracer_amd import Backend, PtmiError, render_scene, structs as S
from opencl_pathtracer_amd import backend as backend_flags

pytestmark = pytest.mark.gpu

RMS_TOL = 1e-4  # north_star: "within 1e-4 per-channel RMS"


KERNELS = {"wavefront": 0, "megakernel": 2}  # PTMI_FLAG_MEGAKERNEL = 2: both kernels must give the same bits


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("case", cases.SMALL)
def test_bit_exact_vs_oracle(case, kernel, scene_factory):
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    spp = 6
    color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, spp, sampler=sampler, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, spp, sampler=sampler)
    assert np.array_equal(count, o_count)
    assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert counters == totals
    bad = np.argwhere(color.view(np.uint32) != o_color.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} differing channel values, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_bit_exact_vs_oracle_1m_triangles(kernel, scene_factory):
    name, sampler, w, h, d = cases.CASES["tris1m_160x90_d10"]
    sc = scene_factory(name, w, h)
    color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, 2, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, 2)
    assert counters == totals
    assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_random_sampler_vs_oracle(kernel, scene_factory):
    sc = scene_factory("cornell", 64, 48)
    color, count, (dep, bbx, tri), counters = render_scene(sc, 64, 48, 4, 8, sampler=S.RANDOM, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 64, 48, 4, 8, sampler=S.RANDOM)
    assert np.array_equal(count, o_count) and counters == totals and np.array_equal(dep, o_dep)
    assert np.allclose(color, o_color, rtol=1e-5, atol=1e-5)
    assert (cases.rms_per_channel(color, count, o_color, o_count) <= 1e-6).all()


def test_iteration_ranges_compose(scene_factory):
    """[0,16) in one launch == [0,8) then [8,16) on the same context, bit for bit (sequential accumulation);
    and two contexts rendering the two shards sum to the same image up to fp32 rounding (spp sharding, 8e)."""
    sc = scene_factory("matmix", 96, 96)
    full, full_n, _, _ = render_scene(sc, 96, 96, 8, 16)
    be = Backend().setup_context(96, 96, 8, sc.lightsSize)
    be.initialize_memory(sc)
    be.render(0, 8)
    be.render(8, 8)
    two, two_n = be.read_image()
    be.release()
    assert np.array_equal(full.view(np.uint32), two.view(np.uint32)) and np.array_equal(full_n, two_n)
    a, an, _, _ = render_scene(sc, 96, 96, 8, 8, first_iteration=0)
    b, bn, _, _ = render_scene(sc, 96, 96, 8, 8, first_iteration=8)
    assert np.array_equal(an + bn, full_n)
    assert (cases.rms_per_channel(a + b, an + bn, full, full_n) <= 1e-6).all()


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_edge_sizes_and_depth_zero(kernel, scene_factory):
    """Image sizes that are not multiples of the 8x8 tile, a 1x1 image, and ray depth 0 (no segment traced)."""
    import copy
    for w, h, d, spp in [(13, 7, 3, 3), (1, 1, 2, 5), (9, 17, 0, 2), (70, 5, 1, 1)]:
        sc = copy.copy(scene_factory("cornell", 64, 48))
        color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, spp, flags=KERNELS[kernel])
        o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, spp)
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)
        assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and counters == totals, (w, h, d)


def _custom_scene(tris, base):
    import copy
    from opencl_pathtracer_amd import bvh_create
    sc = copy.copy(base)
    sc.triangulation = tris
    sc.bvh = None
    return bvh_create(sc)


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_degenerate_trees_and_big_leaves(kernel, scene_factory):
    """Root that is a leaf (<= 4 triangles), and a leaf of 40 coincident triangles (NODE_LEAF_MIN_DIAG): the
    reference puts no bound on a leaf's size; the device layout keeps such leaves in a side table."""
    from opencl_pathtracer_amd import scenes
    base = scenes.cornell_box(48, 32)
    rs = np.random.RandomState(5)
    few = (rs.uniform(100, 450, (3, 1, 3)) + rs.uniform(-120, 120, (3, 3, 3))).astype(np.float32)
    one_leaf = _custom_scene(scenes.triangle_create(few[:, 0], few[:, 1], few[:, 2]), base)
    assert len(one_leaf.bvh) == 1 and one_leaf.bvh["isLeaf"][0]
    tri0 = (np.array([[[150, 200, 100], [420, 230, 120], [260, 300, 420]]], np.float32))
    stack = np.repeat(tri0, 40, axis=0) + rs.uniform(-1e-3, 1e-3, (40, 3, 3)).astype(np.float32)
    wall = scenes.cornell_box(48, 32).triangulation[:10]
    big = scenes.triangle_create(stack[:, 0], stack[:, 1], stack[:, 2])
    both = scenes._concat_tris([wall.copy(), big])
    big_leaf = _custom_scene(both, base)
    assert big_leaf.bvh["nbTriangles"][big_leaf.bvh["isLeaf"] != 0].max() > 6
    for sc in (one_leaf, big_leaf):
        color, count, (dep, bbx, tri), counters = render_scene(sc, 48, 32, 4, 3, flags=KERNELS[kernel])
        o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 48, 32, 4, 3)
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and counters == totals
        assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_leaf_order_with_exact_ties(kernel, scene_factory):
    """Triangles of one leaf at EXACTLY the same distance: the reference's leaf loop accepts a hit whose squared distance
    equals the limit (`> limit` rejects, FullKernel.cl:560), so the LAST of several coincident triangles wins and a shadow
    query stops counting at the FIRST.  A panel stored several times over with a different material each time - a leaf of
    4 triangles, and leaves of 6 and 7 that a leaf pass of the wavefront kernel takes in two parts - shows the winner in
    the image and the stop in the triangle-test histogram."""
    from opencl_pathtracer_amd import scenes
    base = scenes.cornell_box(48, 32)
    walls = base.triangulation[:10]
    quad = np.array([[[150, 180, 120], [430, 200, 130], [400, 260, 400]], [[150, 180, 120], [400, 260, 400], [170, 250, 380]]], np.float32)
    for n_tris, copies in ((2, 2), (1, 6), (1, 7)):
        panels = [scenes.triangle_create(quad[:n_tris, 0], quad[:n_tris, 1], quad[:n_tris, 2], mat_pos=k % 3) for k in range(copies)]
        sc = _custom_scene(scenes._concat_tris([walls.copy()] + panels), base)
        assert sc.bvh["nbTriangles"][sc.bvh["isLeaf"] != 0].max() == n_tris * copies  # the copies share one leaf
        color, count, (dep, bbx, tri), counters = render_scene(sc, 48, 32, 4, 4, flags=KERNELS[kernel])
        o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 48, 32, 4, 4)
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and counters == totals, copies
        assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri), copies


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_russian_roulette_mode_vs_oracle(kernel, scene_factory):
    """PTMI_FLAG_RUSSIAN_ROULETTE: the termination block the reference ships commented out (FullKernel.cl:1306-1314), as
    written there - a non-parity mode (images differ from the reference's), bit-exact against the oracle's same switch."""
    from opencl_pathtracer_amd.backend import FLAG_RUSSIAN_ROULETTE
    w, h, d = 64, 48, 16
    sc = scene_factory("cornell", w, h)
    color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, 6, flags=KERNELS[kernel] | FLAG_RUSSIAN_ROULETTE)
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, 6, russian_roulette=True)
    assert counters == totals and np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)
    plain, _, (p_dep, _, _), _ = render_scene(sc, w, h, d, 6, flags=KERNELS[kernel])
    assert dep[7:].sum() > 0 and dep[d] < p_dep[d]  # paths now end between the 7th bounce and the depth limit
    assert not np.array_equal(color, plain)


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_source_seed_mode_vs_oracle(kernel, scene_factory):
    """PTMI_FLAG_SOURCE_SEED: InitializeRandomSeed's zero test where the source text has it (on the square) - a non-parity mode.
    96 x 96: index = x + 96 y + 9216 it is a multiple of 2^16 for pixel (64, 42) at iteration 28: the compiled reference (and the
    parity modes) keep seed 0 there and draw 0 for every random number; with the flag that path gets seed 1.  Every other pixel of
    the iteration is untouched; bit-exact against the oracle's same switch."""
    w, h, d = 96, 96, 8
    sc = scene_factory("matmix", w, h)
    flags = KERNELS[kernel] | backend_flags.FLAG_SOURCE_SEED
    color, count, _, counters = render_scene(sc, w, h, d, 1, first_iteration=28, flags=flags)
    o_color, o_count, _, totals = O.oracle_render(sc, w, h, d, 1, first_iteration=28, source_seed=True)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count) and counters == totals
    parity, _, _, _ = render_scene(sc, w, h, d, 1, first_iteration=28, flags=KERNELS[kernel])
    differs = np.argwhere((color.view(np.uint32) != parity.view(np.uint32)).any(-1))
    assert differs.tolist() == [[42, 64]]


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_leaf_without_triangles_next_to_a_leaf(kernel, scene_factory):
    """A hand-made tree the product builder never emits: a leaf with nbTriangles == 0 whose box is NOT flagged isEmpty,
    sibling of an ordinary leaf.  The reference descends into it (its leaf loop runs zero times, FullKernel.cl:638-646)
    and pops the sibling; the device layout stores such a leaf as an empty child - same image, same counters - instead of
    letting the wavefront kernel pop a leaf reference where it expects an inner node."""
    import copy
    base = scene_factory("cornell", 64, 48)
    sc = copy.copy(base)
    bvh = base.bvh
    leaf = bvh["isLeaf"] != 0
    # every inner node both of whose children are leaves gets its second child replaced by a new inner node
    # (empty leaf E, the old leaf B), E carrying B's box: rays meet E as the near child or as the far one
    parents = [i for i in range(len(bvh)) if not leaf[i] and leaf[bvh["son1Id"][i]] and leaf[bvh["son2Id"][i]]]
    assert parents
    extra = np.zeros(2 * len(parents), dtype=S.Node)
    new = np.frombuffer(bytearray(bvh.tobytes() + extra.tobytes()), dtype=S.Node)
    for j, p in enumerate(parents):
        y, e = len(bvh) + 2 * j, len(bvh) + 2 * j + 1
        b = int(new["son2Id"][p])
        new[y] = new[b]
        new["isLeaf"][y] = 0
        new["nbTriangles"][y] = 0
        new["cutAxis"][y] = (int(new["cutAxis"][p]) + 1) % 3
        new["son1Id"][y], new["son2Id"][y] = (e, b) if j % 2 == 0 else (b, e)
        new[e] = new[b]
        new["nbTriangles"][e] = 0
        new["trianglesAABB"]["isEmpty"][e] = 0
        new["son2Id"][p] = y
    sc.bvh = new
    sc.bvhMaxDepth = base.bvhMaxDepth + 1
    color, count, (dep, bbx, tri), counters = render_scene(sc, 64, 48, 4, 4, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 64, 48, 4, 4)
    assert counters == totals and counters["box_tests"] > O.oracle_render(base, 64, 48, 4, 4)[3]["box_tests"]
    assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)
    # the extra nodes change the work, not the picture
    plain, _, _, _ = render_scene(base, 64, 48, 4, 4, flags=KERNELS[kernel])
    assert np.array_equal(color.view(np.uint32), plain.view(np.uint32))


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_unconventional_w_components(kernel, scene_factory):
    """OpenCL's dot/normalize are 4-component: scenes whose w lanes break the importer's conventions (points
    w=1, N.w=1, normals w=0) must still equal the reference algorithm, which the oracle evaluates literally."""
    import copy
    sc = copy.copy(scene_factory("matmix", 96, 96))
    t = sc.triangulation.copy()
    rs = np.random.RandomState(11)
    t["N"][:, 3] = rs.uniform(-0.5, 1.5, len(t)).astype(np.float32)
    t["S1"][:, 3] = 1.0 + rs.uniform(-0.05, 0.05, len(t)).astype(np.float32)
    t["S3"][:, 3] = 1.0 + rs.uniform(-0.05, 0.05, len(t)).astype(np.float32)
    t["N2"][:, 3] = rs.uniform(-0.2, 0.2, len(t)).astype(np.float32)
    sc.triangulation = t
    sc.cameraDirection = sc.cameraDirection.copy()
    sc.cameraDirection[3] = 0.05
    color, count, (dep, bbx, tri), counters = render_scene(sc, 96, 96, 8, 3, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 96, 96, 8, 3)
    assert counters == totals and np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx)
    same = color.view(np.uint32) == o_color.view(np.uint32)
    nan_both = np.isnan(color) & np.isnan(o_color)
    assert (same | nan_both).all()


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_axis_aligned_rays_and_unordered_boxes_take_the_literal_box_test(kernel, scene_factory):
    """The wavefront kernel's short slab test (box_hit_ordered) is only valid without NaNs; rays with a zero
    direction component (reciprocal +-inf, and the reference's quirk that a +0 component fails every box while -0
    works, FullKernel.cl:79-83) and trees whose boxes are not pMin <= pMax must take the literal form."""
    import copy
    from opencl_pathtracer_amd import scenes
    base = scene_factory("cornell", 64, 48)
    for direction in ((0.0, 0.0, -1.0), (-0.0, -0.0, -1.0), (0.0, 1.0, 0.0)):
        sc = copy.copy(base)
        sc.lights = scenes._records([scenes.light_directional(direction, power=2.0),
                                     scenes.light_point((278.0, 279.5, 420.0), power=60000.0)], S.Light)
        # shadow rays start on axis-aligned walls: many box planes pass exactly through the origin (0 * inf)
        color, count, (dep, bbx, tri), counters = render_scene(sc, 64, 48, 4, 3, sampler=S.UNIFORM, flags=KERNELS[kernel])
        o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 64, 48, 4, 3, sampler=S.UNIFORM)
        assert counters == totals, direction
        assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))
    # 0 * inf: all camera rays lie in the plane x = 278 (d.x = -0 for a quarter of them, +0 for the rest) and 300
    # small triangles put box planes exactly there, so slab ends are NaN and only the literal comparisons decide
    # (checked once by mutation: with the per-ray check disabled this case fails)
    rs = np.random.RandomState(3)
    c = rs.uniform((150, 50, 50), (400, 500, 500), (300, 3)).astype(np.float32)
    v = (c[:, None, :] + rs.uniform(-40, 40, (300, 3, 3))).astype(np.float32)
    side = rs.randint(0, 2, 300)
    v[:, 0, 0] = 278.0  # first vertex on the plane, the others strictly on one side of it
    v[:, 1:, 0] = np.where(side[:, None] == 1, 278.0 + rs.uniform(1, 60, (300, 2)), 278.0 - rs.uniform(1, 60, (300, 2)))
    extra = scenes.triangle_create(v[:, 0], v[:, 1], v[:, 2])
    sc = _custom_scene(scenes._concat_tris([base.triangulation.copy(), extra]), base)
    sc.cameraPosition = np.array([278.0, -800.0, 273.0, 1.0], np.float32)
    sc.cameraDirection = np.array([-0.0, 1.0, 0.0, 0.0], np.float32)
    sc.cameraRight = np.array([0.0, 0.0, 0.5, 0.0], np.float32)
    sc.cameraUp = np.array([0.0, 0.0, 0.3, 0.0], np.float32)
    color, count, (dep, bbx, tri), counters = render_scene(sc, 64, 48, 4, 3, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 64, 48, 4, 3)
    assert counters == totals and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))
    assert counters["surface_hits"] > 0
    # a hand-made tree with one inverted box: every ray must fall back to the literal test
    sc = copy.copy(scene_factory("tris20k", 96, 64))
    bvh = sc.bvh.copy()
    inner = np.flatnonzero(bvh["isLeaf"] == 0)[5]
    lo, hi = bvh["trianglesAABB"]["pMin"][inner].copy(), bvh["trianglesAABB"]["pMax"][inner].copy()
    bvh["trianglesAABB"]["pMin"][inner, 0], bvh["trianglesAABB"]["pMax"][inner, 0] = hi[0], lo[0]
    sc.bvh = bvh
    color, count, (dep, bbx, tri), counters = render_scene(sc, 96, 64, 6, 2, flags=KERNELS[kernel])
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 96, 64, 6, 2)
    assert counters == totals and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))


def test_deep_paths_keep_the_direct_statistics_atomics(scene_factory):
    """Path depths that do not fit the staged statistics word (>= 64) fall back to the reference's three atomics."""
    sc = scene_factory("matmix", 48, 32)
    color, count, (dep, bbx, tri), counters = render_scene(sc, 48, 32, 70, 2)
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 48, 32, 70, 2)
    assert counters == totals and np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))


def test_display_scanlines_equal_the_host_quantisation(scene_factory):
    """ptmi_read_display (device) == ConvertRGBAToBMPBuffer restated on the host (output.to_bmp_buffer), byte for byte,
    including the padding of a width that is not a multiple of 4, never-sampled pixels (0/0 -> 255) and the
    negative-red marker."""
    from opencl_pathtracer_amd import output
    w, h = 50, 37  # 150 bytes per row -> 2 padding bytes
    sc = scene_factory("matmix", w, h)
    be = Backend().setup_context(w, h, 6, sc.lightsSize, S.JITTERED)
    be.initialize_memory(sc)
    shown = be.read_display()                      # nothing rendered yet: 0/0 everywhere
    assert shown.shape == (h, 152) and (shown[:, :150] == 255).all() and (shown[:, 150:] == 0).all()
    be.render(0, 5)
    color, count = be.read_image()
    expect, stride = output.to_bmp_buffer(color, count)
    assert stride == 152
    assert be.read_display().tobytes() == expect
    # the negative-red marker, saturation and never-sampled pixels, on accumulators loaded with ptmi_write_image
    rs = np.random.RandomState(2)
    c = (rs.uniform(-0.5, 3.0, (h, w, 4))).astype(np.float32)
    n = rs.randint(0, 3, (h, w)).astype(np.float32)
    expect, _ = output.to_bmp_buffer(c, n)
    be.write_image(c, n)
    back_c, back_n = be.read_image()
    assert np.array_equal(back_c, c) and np.array_equal(back_n, n)
    assert be.read_display().tobytes() == expect
    # resuming: 3 iterations on top of loaded accumulators == 5 + 3 in one go
    be.write_image(color, count)
    be.render(5, 3)
    r_color, r_count = be.read_image()
    be.release()
    o_color, o_count, _, _ = O.oracle_render(sc, w, h, 6, 8)
    assert np.array_equal(r_count, o_count) and np.array_equal(r_color.view(np.uint32), o_color.view(np.uint32))


def test_reinitialising_and_releasing_returns_the_device_memory(scene_factory):
    """Every allocation of ptmi_initialize_memory / ptmi_render / ptmi_read_display is released again (a 33 KB block
    once was not): free device memory after 40 set-up / render / release cycles equals the level after the first."""
    import torch  # (device-memory query only: a second, ctypes-loaded HIP runtime may not see the device on every box)
    def free_bytes():
        return torch.cuda.mem_get_info(0)[0]
    sc = scene_factory("tris20k", 64, 48)
    levels = []
    for k in range(40):
        be = Backend().setup_context(64, 48, 4, sc.lightsSize, S.JITTERED)
        be.initialize_memory(sc)
        be.initialize_memory(sc)  # a second scene upload on the same context frees the first
        be.render(0, 2)
        be.read_display()
        be.release()
        levels.append(free_bytes())
    # (from the fifth cycle on: in a fresh process the HIP runtime itself still grows its pools - 8 MiB once - during the first)
    assert levels[-1] >= levels[5] - (1 << 20), (levels[5], levels[-1])  # (allocator granularity: well under 35 x 33 KB)
    assert max(levels[5:]) - min(levels[5:]) <= (2 << 20)


def test_wide_record_addresses(scene_factory, monkeypatch):
    """Record arrays of 4 GB or more need 64-bit byte offsets; the switch forces that path on a small scene."""
    monkeypatch.setenv("PTMI_WIDE_RECORDS", "1")
    sc = scene_factory("tris20k", 96, 64)
    color, count, (dep, bbx, tri), counters = render_scene(sc, 96, 64, 6, 2)
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 96, 64, 6, 2)
    assert counters == totals and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))


def test_plain_and_general_shading_specialisations(scene_factory, monkeypatch):
    """A scene of plain-colour MAT_STANDART materials and LIGHT_POINT lights runs the wavefront kernel's plain-shading
    specialisation; PTMI_GENERIC_SHADING=1 sends the same scene through the general one: both equal the oracle bit for bit
    (the general specialisation is what every scene with textures, other materials or other lights runs anyway)."""
    sc = scene_factory("cornell", 96, 64)
    assert (sc.materiaux["type"] == S.MAT_STANDART).all() and sc.materiaux["isSimpleColor"].all() and (sc.lights["type"] == S.LIGHT_POINT).all()
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, 96, 64, 6, 5)
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("PTMI_GENERIC_SHADING", "1")
        color, count, (dep, bbx, tri), counters = render_scene(sc, 96, 64, 6, 5)
        assert counters == totals and np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri), generic
        assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count), generic


def test_many_iterations_in_one_call_are_chunked(scene_factory):
    """ptmi_render splits a long range into launches of <= 32 iterations (staging array bound): same bits."""
    sc = scene_factory("cornell", 64, 48)
    color, count, (dep, _, _), counters = render_scene(sc, 64, 48, 4, 37)
    o_color, o_count, (o_dep, _, _), totals = O.oracle_render(sc, 64, 48, 4, 37)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and (count == 37).all()
    assert np.array_equal(dep, o_dep) and counters == totals


def test_full_size_properties_1080p_1m_triangles(scene_factory):
    """BASELINE config 3 at full size (1920x1080, 1M triangles, depth 10), where the oracle would take minutes:
    size-independent properties instead -- every pixel sampled exactly spp times, histograms sum to the paths,
    counter identities, finite non-negative radiance, and spp shards of two contexts add up to one render."""
    w, h, d, spp = 1920, 1080, 10, 4
    sc = scene_factory("tris1m", w, h)
    color, count, (dep, bbx, tri), c = render_scene(sc, w, h, d, spp)
    assert (count == spp).all() and np.isfinite(color).all() and (color >= 0).all()
    paths = w * h * spp
    assert dep.sum() == paths == c["paths"] and bbx.sum() <= paths and tri.sum() <= paths
    k = np.arange(d + 1, dtype=np.int64)
    assert (dep.astype(np.int64) * k).sum() == c["surface_hits"] == c["shadow_rays"]  # one light: one shadow ray per hit
    assert c["surface_hits"] <= c["segments"] <= c["surface_hits"] + paths
    k5 = np.arange(5000, dtype=np.int64)
    assert (bbx.astype(np.int64) * k5).sum() <= c["box_tests"] and c["box_tests"] % 2 == 0  # two box tests per node visit
    assert 1500 < c["box_tests"] / paths < 1900 and 400 < c["triangle_tests"] / paths < 600  # SURVEY probe: 1675 / 490
    a, an, (ad, _, _), ac = render_scene(sc, w, h, d, 2, first_iteration=0)
    b, bn, (bd, _, _), bc = render_scene(sc, w, h, d, 2, first_iteration=2)
    assert np.array_equal(ad + bd, dep) and all(ac[key] + bc[key] == c[key] for key in c)
    assert np.array_equal(an + bn, count)
    assert (cases.rms_per_channel(a + b, an + bn, color, count) <= 1e-6).all()


def test_clear_and_reinitialize(scene_factory):
    sc = scene_factory("cornell", 64, 48)
    be = Backend().setup_context(64, 48, 4, 1)
    be.initialize_memory(sc)
    be.render(0, 3)
    first, _ = be.read_image()
    be.clear()
    zero, zero_n = be.read_image()
    assert not zero.any() and not zero_n.any()
    assert be.counters()["paths"] == 0
    be.render(0, 3)
    again, _ = be.read_image()
    assert np.array_equal(first, again)
    be.initialize_memory(sc)  # re-upload on the same context
    be.render(0, 3)
    third, _ = be.read_image()
    be.release()
    assert np.array_equal(first, third)


def test_run_kernel_mirrors_reference_loop(scene_factory):
    """OpenCL_RunKernel's contract: callback once per image after the readback, stats after the loop, release."""
    sc = scene_factory("cornell", 64, 48)
    calls = []
    be = Backend().setup_context(64, 48, 4, 1)
    be.initialize_memory(sc)
    color, count, (dep, _, _), times = be.run_kernel(lambda: calls.append(1) or False, 5)
    assert len(calls) == 5 and (count == 5).all() and dep.sum() == 64 * 48 * 5 and len(times) == 3
    with pytest.raises(PtmiError):
        be.render(0, 1)  # released, like the cl objects at OpenCL.cpp:120-139


def test_histograms_can_be_disabled(scene_factory):
    from opencl_pathtracer_amd.backend import FLAG_NO_HISTOGRAMS
    sc = scene_factory("cornell", 64, 48)
    color, count, (dep, bbx, tri), counters = render_scene(sc, 64, 48, 4, 2, flags=FLAG_NO_HISTOGRAMS)
    ref, _, _, ref_counters = render_scene(sc, 64, 48, 4, 2)
    assert not dep.any() and not bbx.any() and not tri.any()
    assert counters == ref_counters and np.array_equal(color, ref)


def test_bad_scene_is_an_error_not_a_fault(scene_factory):
    sc = scene_factory("cornell", 64, 48)
    import copy
    broken = copy.copy(sc)
    broken.bvh = sc.bvh.copy()
    inner = np.flatnonzero(broken.bvh["isLeaf"] == 0)[0]
    broken.bvh["son2Id"][inner] = 10 ** 6
    be = Backend().setup_context(64, 48, 4, 1)
    with pytest.raises(PtmiError) as e:
        be.initialize_memory(broken)
    assert e.value.code == -5
    broken.bvh = sc.bvh.copy()
    broken.bvh["son2Id"][inner] = inner  # cycle
    with pytest.raises(PtmiError):
        be.initialize_memory(broken)
    broken = copy.copy(sc)
    broken.triangulation = sc.triangulation.copy()
    broken.triangulation["materialWithNegativeNormalIndex"][3] = 99
    with pytest.raises(PtmiError):
        be.initialize_memory(broken)
    with pytest.raises(PtmiError):
        be.render(0, 1)  # no scene resident
    be.release()
    # texture coordinates beyond the int range would index texels outside the texture (header.cl:430-459): refused for textured triangles
    from opencl_pathtracer_amd import scenes, bvh_create
    tex = bvh_create(scenes.build("feat_textured", 64, 64))
    textured = np.flatnonzero(tex.materiaux["isSimpleColor"][tex.triangulation["materialWithPositiveNormalIndex"]] == 0)
    plain = np.flatnonzero(tex.materiaux["isSimpleColor"][tex.triangulation["materialWithPositiveNormalIndex"]] != 0)
    be = Backend().setup_context(64, 64, 4, 1)
    tex.triangulation["UVP2"][plain[0]] = (1e30, np.inf)  # an untextured triangle's coordinates are never read: fine
    be.initialize_memory(tex)
    for bad in (1e30, np.inf, np.nan):
        tex.triangulation["UVN3"][textured[0]] = (0.5, bad)
        with pytest.raises(PtmiError, match="texture coordinates") as e:
            be.initialize_memory(tex)
        assert e.value.code == -5
    be.release()


@pytest.mark.parametrize("case", list(cases.CASES))
def test_vs_reference_kernel_on_gpu(case, scene_factory):
    """The reference's own Kernel_Main (unmodified source -> gfx950 code object, built as its own build line builds it:
    OpenCL default arithmetic) on the same inputs, 64 spp.

      * PTMI_FLAG_DEFAULT_ARITHMETIC: the integrator restates that build's arithmetic - every sum, count and histogram bin
        is EQUAL (RMS 0; the full bit-for-bit suite is tests/test_reference_default_gpu.py);
      * the strict mode (no flag) equals the reference's strict build bit for bit (tests/test_reference_strict_gpu.py), so its
        distance to the default build IS the reference's own strict-vs-default distance: the two RMS figures are the same
        number, asserted with ==, no tolerance."""
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present (built only where the reference tree exists)")
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    spp = 64
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, spp)
    color, count, (dep, bbx, tri), _ = render_scene(sc, w, h, d, spp, sampler=sampler, flags=backend_flags.FLAG_DEFAULT_ARITHMETIC)
    assert np.array_equal(count, r_count)
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    assert np.array_equal(color.view(np.uint32), r_color.view(np.uint32))
    assert cases.rms_per_channel(color, count, r_color, r_count).max() == 0.0
    if O.have_ref_kernel(case, strict=True):
        s_color, s_count, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp, strict=True)
        g_color, g_count, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler)
        floor = cases.rms_per_channel(s_color, s_count, r_color, r_count)
        rms = cases.rms_per_channel(g_color, g_count, r_color, r_count)
        print(f"{case}: {spp} spp: strict mode vs reference default build {rms}, reference strict vs default {floor}")
        assert np.array_equal(rms, floor), (rms, floor)


@pytest.mark.parametrize("flags", [0, backend_flags.FLAG_DEFAULT_ARITHMETIC])
def test_scheduler_statistics_and_the_leaf_pass_item_protocol(flags, scene_factory):
    """PTMI_FLAG_SCHEDULER_STATS builds: same image as the production build, plausible trip counts - and ZERO leaf-pass items
    read with an owner lane or a record index out of range (an item slot read before its final writer, the cause of the two
    faults of round 2's experimental 'blind item writes' variant, would show here first)."""
    name, sampler, w, h, d = cases.CASES["tris1m_160x90_d10"]
    sc = scene_factory(name, w, h)
    be = Backend().setup_context(w, h, d, sc.lightsSize, sampler, flags=flags | backend_flags.FLAG_SCHEDULER_STATS)
    be.initialize_memory(sc)
    be.render(0, 8)
    color, count = be.read_image()
    st, c = be.scheduler_stats(), be.counters()
    be_checks = be.invariant_checks()
    be.release()
    ref_color, ref_count, _, ref_c = render_scene(sc, w, h, d, 8, sampler=sampler, flags=flags)
    assert np.array_equal(color.view(np.uint32), ref_color.view(np.uint32)) and np.array_equal(count, ref_count) and c == ref_c
    assert st["leaf_item_violations"] == 0
    # the reference's -D LOG_INFO device-side checks (header.cl:21-48), counted by this build: a clean render
    assert be_checks == {"sample_out_of_range": 0, "normal_not_facing_ray": 0, "negative_direct_radiance": 0,
                         "scattered_below_surface": 0, "statistics_out_of_range": 0, "refraction_undefined_in_reference": 0}
    assert st["trips_node"] > 0 and st["trips_triangle"] > 0 and st["trips_path"] > 0
    # every counted triangle test was one item of one pass (a shadow query stops COUNTING at its first hit: items dealt out
    # behind it in the same pass are tested and not counted)
    assert c["triangle_tests"] <= st["lanes_triangle"] <= 1.2 * c["triangle_tests"]
    assert 0 < st["cycles_path"] < st["cycles_loop"]


# (case, samples per pixel): each BASELINE config's own sample count on its parity-size scene - config 2 (Cornell box)
# 1024 spp, config 3 (1M triangles) 256 spp, config 5's stand-in (material mix) 2048 spp
NORTH_STAR = [("cornell_64x48_d4", 1024), ("cornell_128x128_d8", 1024), ("tris20k_96x64_d6", 256),
              ("tris1m_160x90_d10", 256), ("matmix_96x96_d8", 2048)]


@pytest.mark.parametrize("case,spp", NORTH_STAR)
def test_north_star_rms_at_config_spp(case, spp, scene_factory):
    """north_star: "image matches the reference OpenCL kernel on the same scene/seed within 1e-4 per-channel RMS", asserted
    where it is quoted - at each config's own sample count, against the kernel the reference's own build line produces -
    and literally: `<= 1e-4`, no escape.  In the default-arithmetic mode the images are in fact equal (RMS 0).  The strict
    mode's distance to that build (= the reference's own strict-vs-default distance) is printed and recorded beside it
    (profiles/r03_north_star_rms.json is a copy of what this test writes on the GPU box)."""
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    r_color, r_count, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp)
    color, count, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler, flags=backend_flags.FLAG_DEFAULT_ARITHMETIC)
    assert np.array_equal(count, r_count)
    rms = cases.rms_per_channel(color, count, r_color, r_count)
    g_color, g_count, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler)
    strict = cases.rms_per_channel(g_color, g_count, r_color, r_count)
    print(f"{case} at {spp} spp: per-channel rms vs the reference's default build: default-arithmetic mode {rms}, strict mode {strict}")
    _record_north_star(case, spp, rms, strict)
    assert (rms <= RMS_TOL).all(), rms
    assert np.array_equal(color.view(np.uint32), r_color.view(np.uint32))


def _record_north_star(case, spp, rms, floor):
    import json
    import os
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if not os.path.isdir(out_dir):
        return
    path = os.path.join(out_dir, "r03_north_star_rms.json")
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[case] = {"spp": spp, "rms_default_arithmetic_mode_vs_reference_default_build": [float(x) for x in rms],
                  "rms_strict_mode_vs_reference_default_build": None if floor is None else [float(x) for x in floor]}
    json.dump(data, open(path, "w"), indent=1)


@pytest.mark.parametrize("seed", [3, 6, 12, 21, 41, 43])
def test_fuzzed_scenes_in_every_other_mode_vs_oracle(seed):
    """The seeded scenes of scenes.fuzz_scene (odd seeds with the hostile records, seeds from 40 with corrupted ones) through
    what the reference-kernel fuzz tests do not reach: the RANDOM sampler, Russian roulette, the source-seed rule, the
    statistics build of the wavefront kernel (same image, no item-protocol violation - or, for a scene served by the
    one-path-per-lane kernel, no statistics), and two listed devices.  Bit-exact against the oracle in both arithmetics."""
    import warnings
    from opencl_pathtracer_amd import scenes, bvh_create
    w, h, d = 72, 40, 7
    name = f"fuzz{seed}{'h' if seed % 2 else ''}{'r' if seed >= 40 else ''}_l{1 + 2 * (seed % 3 == 0)}"
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = bvh_create(scenes.build(name, w, h))
    DA = backend_flags.FLAG_DEFAULT_ARITHMETIC
    for da in (False, True):
        f = DA if da else 0
        base = render_scene(sc, w, h, d, 5, flags=f)
        o = O.oracle_render(sc, w, h, d, 5, default_arithmetic=da)
        assert np.array_equal(base[0].view(np.uint32), o[0].view(np.uint32)) and np.array_equal(base[1], o[1]) and base[3] == o[3]
        assert all(np.array_equal(a, b) for a, b in zip(base[2], o[2]))
        # RANDOM sampler: atomic float sums, so equal counts and close colours (NaN where the oracle has NaN)
        color, count, (dep, _, _), counters = render_scene(sc, w, h, d, 5, sampler=S.RANDOM, flags=f)
        o_color, o_count, (o_dep, _, _), totals = O.oracle_render(sc, w, h, d, 5, sampler=S.RANDOM, default_arithmetic=da)
        assert np.array_equal(count, o_count) and counters == totals and np.array_equal(dep, o_dep)
        assert np.allclose(color, o_color, rtol=1e-4, atol=1e-5, equal_nan=True)
        # Russian roulette and the source-seed rule (non-parity modes)
        rr = render_scene(sc, w, h, d, 5, flags=f | backend_flags.FLAG_RUSSIAN_ROULETTE | backend_flags.FLAG_SOURCE_SEED)
        o_rr = O.oracle_render(sc, w, h, d, 5, russian_roulette=True, source_seed=True, default_arithmetic=da)
        assert np.array_equal(rr[0].view(np.uint32), o_rr[0].view(np.uint32)) and rr[3] == o_rr[3]
        # the statistics build
        be = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, flags=f | backend_flags.FLAG_SCHEDULER_STATS)
        try:
            be.initialize_memory(sc)
            literal = be.literal_kernel_reason() is not None
            be.render(0, 5)
            color, count = be.read_image()
            st = be.scheduler_stats()
        finally:
            be.release()
        assert literal == (seed % 2 == 1)
        assert np.array_equal(color.view(np.uint32), base[0].view(np.uint32)) and np.array_equal(count, base[1])
        # (round 4: a scene whose records yield NaN distances runs the wavefront kernel too - its NANSAFE instantiation)
        assert st["leaf_item_violations"] == 0 and st["trips_node"] > 0
        # one GPU listed twice: sample counts, counters and histograms of the single context; the image up to the association
        # of the two partial sums (NaN pixels of a hostile scene stay NaN)
        two = render_scene(sc, w, h, d, 5, flags=f, devices=[0, 0])
        assert np.array_equal(two[1], base[1]) and two[3] == base[3] and all(np.array_equal(a, b) for a, b in zip(two[2], base[2]))
        assert np.allclose(two[0], base[0], rtol=2e-6, atol=1e-6, equal_nan=True)
