"""The strongest pin of the numerics contract (GPU): the integrator equals the reference kernel BIT FOR BIT.

The reference's arithmetic is implementation-defined where OpenCL leaves it open.  Compiled for this GPU it meets the ROCm
device library: sin()/cos() = __ocml_sin_f32/__ocml_cos_f32, normalize() = v * v_rsq_f32(dot) (a hardware approximation),
dot()/cross() = fma chains.  The integrator and the CPU oracle use exactly those definitions (include/ptmi_detmath.h,
csrc/ptmi_device.hpp, oracle/pt_oracle.c), and every other operation is one correctly rounded IEEE operation - which is
what the reference's STRICT build (-ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt, oracle/_ref/*.strict.hsaco) does
with the unmodified source.  So: image, sample counts and the three histograms of this integrator == the strict reference
build's, exactly; and the oracle == the integrator (tests/test_parity_gpu.py), which pins the oracle to the reference.
Against the reference's DEFAULT build (FMA contraction, approximate divide / sqrt) the distance is then the reference's own
distance to itself.
"""
import ctypes as C
import os

import numpy as np
import pytest

import cases
import oracle_ffi as O
from opencl_pathtracer_amd import render_scene, scenes, bvh_create, structs as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "oracle", "build", "libdevice_math_probe.so")


def test_device_math_is_the_platform_library():
    """ptmi_sincosf (device build) == the device library's sinf/cosf on [0, 2 pi] and beyond; the oracle's table-driven
    v_rsq_f32 == the instruction, for inputs over the whole normal range."""
    if not os.path.exists(PROBE):
        pytest.fail("oracle/build/libdevice_math_probe.so not built (make -C oracle probe)", pytrace=False)
    lib = C.CDLL(PROBE)
    rs = np.random.RandomState(11)
    theta = np.concatenate([np.linspace(0, 2 * np.pi, 200001), rs.uniform(0, 2 * np.pi, 300000), rs.uniform(-50, 50, 100000),
                            np.arange(0, 8.25, 0.25) * np.pi / 4]).astype(np.float32)
    expo = rs.randint(-120, 120, 400000)
    pos = (rs.uniform(1, 2, 400000) * np.exp2(expo.astype(np.float64))).astype(np.float32)
    near_one = rs.uniform(0.25, 4.0, 400000).astype(np.float32)  # squared lengths of what the integrator normalises
    x = np.ascontiguousarray(np.concatenate([theta, pos, near_one]))
    n = len(x)
    outs = [np.empty(n, np.float32) for _ in range(5)]
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    lib.device_math_probe.argtypes = [C.c_void_p, C.c_uint] + [C.c_void_p] * 5
    assert lib.device_math_probe(vp(x), n, *[vp(o) for o in outs]) == 0
    lib_sin, lib_cos, port_sin, port_cos, hw_rsq = outs
    t = slice(0, len(theta))
    assert np.array_equal(lib_sin[t].view(np.uint32), port_sin[t].view(np.uint32))
    assert np.array_equal(lib_cos[t].view(np.uint32), port_cos[t].view(np.uint32))
    # ... and the host build of the same header (what the oracle calls) gives those bits too
    ol = O.oracle()
    s, c = C.c_float(), C.c_float()
    for i in rs.randint(0, len(theta), 20000):
        ol.pto_sincos(float(x[i]), C.byref(s), C.byref(c))
        assert np.float32(s.value).view(np.uint32) == lib_sin[i].view(np.uint32), (x[i], s.value, lib_sin[i])
        assert np.float32(c.value).view(np.uint32) == lib_cos[i].view(np.uint32)
    p = np.arange(len(theta), n)
    emu = np.array([ol.pto_hardware_rsq(float(v)) for v in x[p[::8]]], np.float32)
    assert np.array_equal(emu.view(np.uint32), hw_rsq[p[::8]].view(np.uint32))


STRICT_CASES = [c for c in cases.CASES]


@pytest.mark.parametrize("case", STRICT_CASES)
def test_bit_exact_vs_reference_strict_build(case, scene_factory):
    if not O.have_ref_kernel(case, strict=True):
        O.missing_reference("oracle/_ref strict code object not present (built only where the reference tree exists)")
    name, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(name, w, h)
    spp = 16
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, spp, strict=True)
    color, count, (dep, bbx, tri), _ = render_scene(sc, w, h, d, spp, sampler=sampler)
    assert np.array_equal(count, r_count)
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    bad = np.argwhere(color.view(np.uint32) != r_color.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} channel values differ from the reference's strict build, first at {bad[:5].tolist()}"


FULL_SIZE = {"tris1m_1920x1080_d10": ("tris1m", S.JITTERED, 1920, 1080, 10, 2),   # BASELINE configs[2]: the bench workload
             "cornell_1920x1080_d8": ("cornell", S.JITTERED, 1920, 1080, 8, 4)}     # BASELINE configs[1]


@pytest.mark.parametrize("case", list(FULL_SIZE))
def test_full_size_bit_exact_vs_reference_strict_build(case):
    """BASELINE's own image size, scene and ray depth: every one of the 2 M pixels, the sample counts and the three
    histograms of the default (wavefront) kernel equal the reference kernel's strict build run beside it on this GPU."""
    if not O.have_ref_kernel(case, strict=True):
        O.missing_reference("oracle/_ref strict code object not present (built only where the reference tree exists)")
    name, sampler, w, h, d, spp = FULL_SIZE[case]
    sc = bvh_create(scenes.build(name, w, h))
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, spp, strict=True)
    color, count, (dep, bbx, tri), _ = render_scene(sc, w, h, d, spp, sampler=sampler)
    assert np.array_equal(count, r_count)
    assert np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    bad = np.argwhere(color.view(np.uint32) != r_color.view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} channel values differ from the reference's strict build, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("feature", scenes.FEATURES)
def test_every_feature_bit_exact_vs_reference_strict_build(feature):
    """One kernel feature per scene (materials, textures, light types, sky, two-sided / fallback normals), 256 spp: 1M paths
    each, all five material branches, every light type, and the pixels whose seeds are degenerate."""
    case, w, h, d = "feat_64x64_d8", 64, 64, 8
    if not O.have_ref_kernel(case, strict=True):
        O.missing_reference("oracle/_ref strict code object not present")
    sc = bvh_create(scenes.build("feat_" + feature, w, h))
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, w, h, d, 256, strict=True)
    color, count, (dep, bbx, tri), _ = render_scene(sc, w, h, d, 256)
    assert np.array_equal(count, r_count) and np.array_equal(dep, r_dep)
    assert np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    assert np.array_equal(color.view(np.uint32), r_color.view(np.uint32))


def test_zero_seed_paths_follow_the_compiled_reference(scene_factory):
    """InitializeRandomSeed squares a signed int; where the square wraps to 0 (pixel/iteration index a multiple of 2^16) the
    compiled reference keeps seed 0 - every random number of the path is 0 - instead of the 1 its source suggests
    (oracle/pt_oracle.c).  96 x 96: index = x + 96 y + 9216 it is a multiple of 65536 for (64, 42) at iteration 28."""
    case = "matmix_96x96_d8"
    if not O.have_ref_kernel(case, strict=True):
        # the behaviour is tied to what one compiler makes of an undefined overflow (verified: the image's clang 22 / ROCm 7.2, both
        # builds of the reference kernel): where it cannot be re-verified the test must say so loudly, not pass by skipping
        pytest.fail("oracle/_ref strict code object not present: the seed-0 behaviour of the compiled reference cannot be verified")
    name, sampler, w, h, d = cases.CASES[case]
    assert (64 + 96 * 42 + 9216 * 28) % 65536 == 0
    sc = scene_factory(name, w, h)
    r, _, _, _ = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=28, strict=True)
    g, _, _, _ = render_scene(sc, w, h, d, 1, first_iteration=28)
    o, _, _, _ = O.oracle_render(sc, w, h, d, 1, first_iteration=28)
    assert np.array_equal(g.view(np.uint32), r.view(np.uint32)) and np.array_equal(o.view(np.uint32), g.view(np.uint32))
    # ... and the same in the arithmetic of the reference's own build
    r2, _, _, _ = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=28)
    g2, _, _, _ = render_scene(sc, w, h, d, 1, first_iteration=28, flags=16)
    assert np.array_equal(g2.view(np.uint32), r2.view(np.uint32))
    bounces, _ = O.oracle_trace(sc, w, h, d, 64, 42, 28)
    assert bounces and all(b.seed_after == 0 for b in bounces)
