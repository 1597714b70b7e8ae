"""Known-answer tests for the CPU oracle's building blocks, written independently of it
(python ints / numpy float32), each against the reference lines it restates."""
import ctypes as C

import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import scenes, structs as S

f32 = np.float32


def _arr4(v):
    return (C.c_float * 4)(*[float(x) for x in v])


def py_seed(x, y, w, h, it):
    # header.cl:255-264, int32 wrap-around
    # the zero test is on the un-squared index (what LLVM makes of the signed square, see pt_oracle.c)
    index = (x + y * w + it * w * h) & 0xFFFFFFFF
    s = (index * 2011) & 0xFFFFFFFF
    s = (s * s) & 0xFFFFFFFF
    s = s if index else 1
    return s - (1 << 32) if s & 0x80000000 else s


def py_random(seed):
    # header.cl:246-253: ulong product of a sign-extended int, masked (not modulo) with 0x7FFFFFFF
    seed = (16807 * (seed & 0xFFFFFFFFFFFFFFFF)) & 0x7FFFFFFF
    return seed, f32(seed) / f32(0x7FFFFFFF)


def test_rng_known_answers(built):
    lib = O.oracle()
    for (x, y, w, h, it) in [(0, 0, 64, 48, 0), (1, 0, 64, 48, 0), (63, 47, 64, 48, 7), (1919, 1079, 1920, 1080, 255),
                             (5, 9, 1920, 1080, 4095), (0, 0, 1, 1, 0), (64, 42, 96, 96, 28), (0, 0, 256, 256, 1)]:
        s = lib.pto_initialize_random_seed(x, y, w, h, it)
        assert s == py_seed(x, y, w, h, it)
        cs, ps = C.c_int32(s), s
        for _ in range(8):
            r = lib.pto_random(C.byref(cs))
            ps, pr = py_random(ps)
            assert cs.value == ps and f32(r) == pr and 0.0 <= r <= 1.0
    # seed 0 is remapped to 1; the first draw of seed 1 is 16807 / 2^31
    assert py_seed(0, 0, 1, 1, 0) == 1
    cs = C.c_int32(1)
    assert lib.pto_random(C.byref(cs)) == f32(16807) / f32(2147483648.0) and cs.value == 16807
    # the mask keeps even seeds even forever (Appendix A.6 of SURVEY.md)
    cs = C.c_int32(2 * 12345)
    for _ in range(16):
        lib.pto_random(C.byref(cs))
        assert cs.value % 2 == 0


def test_samplers(built):
    lib = O.oracle()
    out = (C.c_float * 2)()
    # JITTERED (cl:1145-1146): inside the pixel's [0.05, 0.95] sub-square, two draws
    for (x, y, it) in [(0, 0, 0), (10, 20, 3), (63, 47, 9)]:
        s0 = py_seed(x, y, 64, 48, it)
        cs = C.c_int32(s0)
        lib.pto_sampler(S.JITTERED, x, y, 64, 48, it, C.byref(cs), out)
        s1, r1 = py_random(s0)
        s2, r2 = py_random(s1)
        ex = (f32(x) + f32(0.9) * r1 + f32(0.05)) / f32(64) - f32(0.5)
        ey = (f32(y) + f32(0.9) * r2 + f32(0.05)) / f32(48) - f32(0.5)
        assert cs.value == s2 and f32(out[0]) == ex and f32(out[1]) == ey
        assert (x + 0.05) / 64 - 0.5 - 1e-6 <= out[0] <= (x + 0.95) / 64 - 0.5 + 1e-6
    # UNIFORM (cl:1123-1135): 3x3 grid by iteration % 9, no random draw
    for it in range(11):
        cs = C.c_int32(777)
        lib.pto_sampler(S.UNIFORM, 5, 7, 64, 48, it, C.byref(cs), out)
        k = it % 9
        ex = (f32(5) + (f32(k % 3) + f32(0.5)) / f32(3)) / f32(64) - f32(0.5)
        ey = (f32(7) + (f32(k // 3) + f32(0.5)) / f32(3)) / f32(48) - f32(0.5)
        assert cs.value == 777 and f32(out[0]) == ex and f32(out[1]) == ey
    # RANDOM (cl:1137-1141): anywhere in [-0.45, 0.45]^2
    cs = C.c_int32(4242)
    lib.pto_sampler(S.RANDOM, 0, 0, 64, 48, 0, C.byref(cs), out)
    s1, r1 = py_random(4242)
    assert f32(out[0]) == r1 * f32(0.9) + f32(0.05) - f32(0.5)


def test_deterministic_sincos_accuracy(built):
    """ptmi_sincosf: the one free choice OpenCL leaves open (4 ulp allowed); ours stays within 2 ulp on [0, 2pi]."""
    lib = O.oracle()
    s, c = C.c_float(), C.c_float()
    xs = np.concatenate([np.linspace(0, 2 * np.pi, 20001), [0.0, np.pi / 4, np.pi / 2, np.pi, 3 * np.pi / 2]]).astype(f32)
    worst = 0.0
    for x in xs:
        lib.pto_sincos(float(x), C.byref(s), C.byref(c))
        for got, want in ((s.value, np.sin(np.float64(x))), (c.value, np.cos(np.float64(x)))):
            ulp = np.spacing(f32(max(abs(want), 2.0 ** -20)))
            worst = max(worst, abs(got - want) / ulp)
    assert worst <= 2.0, worst
    lib.pto_sincos(0.0, C.byref(s), C.byref(c))
    assert s.value == 0.0 and c.value == 1.0


def test_concentric_disk_and_hemisphere(built):
    lib = O.oracle()
    dx, dy = C.c_float(), C.c_float()
    out = (C.c_float * 4)()
    n = _arr4(np.array([0.3, -0.5, 0.81, 0.0]) / np.linalg.norm([0.3, -0.5, 0.81]))
    for seed in range(1, 4000, 37):
        cs = C.c_int32(seed)
        lib.pto_concentric_sample_disk(C.byref(cs), C.byref(dx), C.byref(dy))
        assert dx.value ** 2 + dy.value ** 2 < 1.0  # the reference's ASSERT, cl:415
        cs = C.c_int32(seed)
        lib.pto_cosine_sample_hemisphere(C.byref(cs), n, out)
        v = np.array(out[:])
        assert abs(np.linalg.norm(v) - 1) < 1e-5 and v[3] == 0 and np.dot(v[:3], np.array(n[:3])) >= -1e-6
    # shortcut |N.z| > 0.9999 (cl:317-320): no rotation
    cs = C.c_int32(99)
    lib.pto_cosine_sample_hemisphere(C.byref(cs), _arr4([0, 0, 1, 0]), out)
    up = np.array(out[:])
    cs = C.c_int32(99)
    lib.pto_cosine_sample_hemisphere(C.byref(cs), _arr4([0, 0, -1, 0]), out)
    assert np.array_equal(np.array(out[:]), -up) and up[2] >= 0


def _box(lo, hi, empty=0):
    b = np.zeros((), S.BoundingBox)
    b["pMin"], b["pMax"], b["isEmpty"] = list(lo) + [1], list(hi) + [1], empty
    return b


def test_bounding_box_quirks(built):
    lib = O.oracle()

    def hit(box, o, d, lim):
        return lib.pto_bounding_box_intersects(box.ctypes.data_as(C.c_void_p), _arr4(o), _arr4(d), f32(lim))

    b = _box((1, -1, -1), (2, 1, 1))
    D = (1, 1e-3, 1e-3, 0)  # slightly off axis, see the +0 quirk below
    assert hit(b, (0, 0, 0, 1), D, np.inf) == 1
    assert hit(b, (0, 0, 0, 1), (-1, 1e-3, 1e-3, 0), np.inf) == 0  # behind
    assert hit(b, (0, 3, 0, 1), D, np.inf) == 0   # misses in y
    assert hit(b, (1.5, 0, 0, 1), (1e-3, 1, 1e-3, 0), 0.0) == 1    # origin inside: true whatever the limit (cl:132)
    assert hit(_box((1, -1, -1), (2, 1, 1), empty=1), (0, 0, 0, 1), D, np.inf) == 0
    # the cull compares the LINEAR entry distance with the SQUARED hit distance (cl:135):
    # entry t = 1; a current hit at distance 0.9 has squared distance 0.81 -> box culled (correct) ...
    assert hit(b, (0, 0, 0, 1), D, 0.81) == 0
    # ... but a hit at distance 3 (squared 9) keeps a box entered at t = 5 alive: 5 < 9
    far = _box((5, -1, -1), (6, 1, 1))
    assert hit(far, (0, 0, 0, 1), D, 9.0) == 1
    # and for distances below 1 a NEARER box is dropped: entry t = 0.5 > 0.6^2 = 0.36
    near = _box((0.5, -1, -1), (0.55, 1, 1))
    assert hit(near, (0, 0, 0, 1), D, 0.36) == 0
    # quirk: a direction component that is exactly +0 takes the "else" slab (cl:79-83,93-97) with inverse = +inf:
    # tyMin = +inf, tyMax = -inf, so "tMin > tyMax" rejects EVERY box; with -0 the slabs come out right
    assert hit(b, (0, 0, 0, 1), (1, 0.0, 1e-3, 0), np.inf) == 0
    assert hit(b, (0, 0, 0, 1), (1, -0.0, -0.0, 0), np.inf) == 1


def test_triangle_intersection(built):
    lib = O.oracle()
    tri = scenes.triangle_create([[0, 0, 0]], [[1, 0, 0]], [[0, 1, 0]])
    s, t, lim = C.c_float(), C.c_float(), C.c_float(np.inf)
    p = (C.c_float * 4)()

    def hit(o, d, limit=np.inf):
        lim.value = limit
        return lib.pto_triangle_intersects(tri.ctypes.data_as(C.c_void_p), _arr4(o), _arr4(d), C.byref(lim),
                                           C.byref(s), C.byref(t), p)

    assert hit((0.25, 0.25, 1, 1), (0, 0, -1, 0)) == 1
    assert abs(lim.value - 1.0) < 1e-6 and abs(p[2]) < 1e-6 and abs(p[3] - 1) < 1e-6
    # barycentrics follow the lexicographically sorted vertices: S1=(0,0,0) S2=(0,1,0) S3=(1,0,0)
    assert np.array_equal(tri["S2"][0][:3], [0, 1, 0]) and abs(s.value - 0.25) < 1e-6 and abs(t.value - 0.25) < 1e-6
    assert hit((0.25, 0.25, -1, 1), (0, 0, 1, 0)) == 1        # two-sided
    assert hit((0.8, 0.8, 1, 1), (0, 0, -1, 0)) == 0          # outside (s+t>1)
    assert hit((0.25, 0.25, 1, 1), (0, 0, 1, 0)) == 0         # behind the origin (cl:564)
    assert hit((0.25, 0.25, 1, 1), (1, 0, 0, 0)) == 0         # parallel (cl:535)
    assert hit((0.25, 0.25, 1, 1), (0, 0, -1, 0), 0.5) == 0   # farther than the current hit (cl:543)
    assert hit((0.25, 0.25, 0.001, 1), (0, 0, -1, 0)) == 0    # closer than sqrt(1e-5): self-hit guard (cl:545)
    assert tri["N"][0][3] == 1.0 and tri["S1"][0][3] == 1.0   # w conventions (host cross gives w=1)


def test_fresnel_and_sky(built):
    lib = O.oracle()
    n = _arr4([0, 0, 1, 0])
    # normal incidence on glass n=1.55: ((n-1)/(n+1))^2
    assert abs(lib.pto_fresnel_glass(_arr4([0, 0, -1, 0]), n) - ((1.55 - 1) / (1.55 + 1)) ** 2) < 1e-6
    assert abs(lib.pto_fresnel_varnish(_arr4([0, 0, -1, 0]), n) - 0.25) < 1e-6
    g = np.array([np.sin(1.5), 0, -np.cos(1.5), 0])
    assert 0.55 < lib.pto_fresnel_glass(_arr4(g), n) <= 1.0  # grazing: towards 1
    sc = scenes.material_mix(8, 8)
    out = (C.c_float * 4)()
    sky = np.ascontiguousarray(sc.sky)
    tex = np.ascontiguousarray(sc.texturesData)
    faces = {}
    for name, d in dict(px=(1, 0.1, 0.2), nx=(-1, 0.1, 0.2), py=(0.1, 1, 0.2), ny=(0.1, -1, 0.2), pz=(0.1, 0.2, 1),
                        nz=(0.1, 0.2, -1)).items():
        lib.pto_sky_color(sky.ctypes.data_as(C.c_void_p), tex.ctypes.data_as(C.c_void_p), _arr4(list(d) + [0]), out)
        faces[name] = tuple(out[:])
        assert out[3] == 0.0  # alpha 255 -> w = 1 - 1
    assert len(set(faces.values())) == 6  # six different faces were hit
    # tie (all components equal) falls through to face 0, uv (0,0) (cl:445-446)
    lib.pto_sky_color(sky.ctypes.data_as(C.c_void_p), tex.ctypes.data_as(C.c_void_p), _arr4([0, 0, 0, 0]), out)
    t0 = sc.texturesData[sc.sky["skyTextures"][0]["offset"]]
    assert np.allclose(out[:3], t0[:3] / 255.0)


def test_render_properties(built, scene_factory):
    """Size-independent properties of a render: counts, histogram sums, thread invariance, shard additivity."""
    sc = scene_factory("matmix", 96, 96)
    color, count, (dep, bbx, tri), tot = O.oracle_render(sc, 96, 96, 8, 4, n_threads=8)
    assert (count == 4).all() and dep.sum() == tot["paths"] == 96 * 96 * 4
    assert (dep * np.arange(9)).sum() == tot["surface_hits"] and tot["shadow_rays"] == 3 * tot["surface_hits"]
    assert tot["segments"] >= tot["surface_hits"] and tot["segments"] <= tot["surface_hits"] + tot["paths"]
    assert (bbx * np.arange(5000)).sum() <= tot["box_tests"] and np.isfinite(color).all() and (color >= 0).all()
    c1, n1, (d1, b1, t1), _ = O.oracle_render(sc, 96, 96, 8, 4, n_threads=1)
    assert np.array_equal(c1, color) and np.array_equal(d1, dep) and np.array_equal(b1, bbx) and np.array_equal(t1, tri)
    # iteration shards: [0,2) then [2,4) accumulated in order == [0,4) bit for bit
    part = O.oracle_render(sc, 96, 96, 8, 2, first_iteration=0)
    part = O.oracle_render(sc, 96, 96, 8, 2, first_iteration=2, into=part)
    assert np.array_equal(part[0], color) and np.array_equal(part[2][0], dep)
    # depth 0: no segment is traced, radiance 0, every path lands in bin 0
    c0, n0, (d0, _, _), t0 = O.oracle_render(sc, 96, 96, 0, 1)
    assert not c0.any() and (n0 == 1).all() and d0[0] == 96 * 96 and t0["segments"] == 0


def test_russian_roulette_switch_of_the_oracle(scene_factory):
    """cl:1306-1314 (commented out in the reference): off by default - the parity mode - and, when on, paths end from the
    7th bounce on with the transfer function divided by the roulette coefficient."""
    sc = scene_factory("cornell", 48, 32)
    off, _, (d_off, _, _), t_off = O.oracle_render(sc, 48, 32, 16, 4)
    on, _, (d_on, _, _), t_on = O.oracle_render(sc, 48, 32, 16, 4, russian_roulette=True)
    assert d_off.sum() == d_on.sum() == 48 * 32 * 4
    assert np.array_equal(d_off[:6], d_on[:6])          # nothing changes for paths of up to five bounces ...
    assert d_on[6:16].sum() > d_off[6:16].sum() and d_on[16] < d_off[16]   # ... then paths end early
    assert t_on["segments"] < t_off["segments"] and np.isfinite(on).all() and not np.array_equal(on, off)
