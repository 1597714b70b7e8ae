"""Pins the CPU oracle to outputs of the REFERENCE ITSELF (no GPU needed here).

tests/golden/ref_<case>.npz were produced on an MI355X by tests/golden/make_reference_fixtures.py from
oracle/_ref/ref_kernel_<case>.hsaco = the reference's Kernel/PathTracer_FullKernel.cl compiled unmodified for
gfx950, in two builds.  Against the STRICT build (IEEE operations in the written order + the platform library's dot /
normalize / sin / cos, which is the oracle's own arithmetic) the comparison is bit for bit
(test_oracle_equals_the_reference_strict_build*).  Against the DEFAULT build (FMA contraction, approximate divide and
sqrt: other legal arithmetic) it is statistical, calibrated on the fixture's own `noise_floor_rms_16spp` = the distance
between the reference's two builds:
  * sample counts exact; depth histograms within 1e-3 of the paths; traversal-work histograms close;
  * per sample (1 spp): <= 1 % flipped, median relative difference <= 1e-6, no bias;
  * 16-spp image: RMS <= max(2e-4, 3 x noise floor), or else <= 0.5 % of the pixels hold a sample that took
    another branch and the rest agree to 1e-4 RMS; both 8-spp shards are checked (spp sharding).
"""
import os

import numpy as np
import pytest

import cases
import oracle_ffi as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _expand(idx, val):
    h = np.zeros(5000, np.int64)
    h[idx] = val
    return h


@pytest.mark.parametrize("case", list(cases.CASES))
def test_oracle_matches_reference_fixture(case, scene_factory):
    path = os.path.join(GOLDEN, f"ref_{case}.npz")
    assert os.path.exists(path), "fixture missing"
    fx = np.load(path)
    name, sampler, w, h, d = cases.CASES[case]
    if name == "tris1m" and os.environ.get("PTMI_SKIP_SLOW"):
        pytest.skip("slow")
    sc = scene_factory(name, w, h)

    import importlib.util
    spec = importlib.util.spec_from_file_location("mkfx", os.path.join(GOLDEN, "make_reference_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    assert str(fx["scene_digest"]) == mk.scene_digest(sc), "scene generator changed since the fixture was made"

    one, _, _, _ = O.oracle_render(sc, w, h, d, 1, sampler=sampler)
    agree = cases.sample_agreement(one, fx["it0_1_color"])
    assert agree["flip_fraction"] <= cases.MAX_FLIP_FRACTION and agree["median_rel"] <= 1e-6, agree

    total_c, total_n, ref_c, ref_n = 0, 0, 0, 0
    work = np.zeros(2, np.int64)
    for first, n in cases.FIXTURE_RANGES:
        tag = f"it{first}_{n}"
        color, count, (dep, bbx, tri), tot = O.oracle_render(sc, w, h, d, n, first_iteration=first, sampler=sampler)
        assert np.array_equal(count, fx[tag + "_count"])
        r_dep = fx[tag + "_depths"].astype(np.int64)
        assert r_dep.sum() == dep.sum() == w * h * n
        assert np.abs(dep.astype(np.int64) - r_dep).sum() <= max(2, 1e-3 * dep.sum())
        # traversal work (sum of the reference's own per-path counters).  The reference's DEFAULT build (FMA
        # contraction, fast divide) does 1.6 % / 3.6 % fewer box / triangle tests than its own strict-IEEE build
        # on the same paths (measured, tools/diag_work.py); the oracle follows IEEE arithmetic, so the loose
        # bound is against the default-build fixture and the tight one (below) against the strict build.
        r_bbx, r_tri = _expand(fx[tag + "_bbx_idx"], fx[tag + "_bbx_val"]), _expand(fx[tag + "_tri_idx"], fx[tag + "_tri_val"])
        k = np.arange(5000)
        for mine, ref in ((bbx, r_bbx), (tri, r_tri)):
            assert abs((mine * k).sum() - (ref * k).sum()) <= 5e-2 * (ref * k).sum()
        work += np.array([(bbx.astype(np.int64) * k).sum(), (tri.astype(np.int64) * k).sum()])
        total_c, total_n = total_c + color, total_n + count
        ref_c, ref_n = ref_c + fx[tag + "_color"], ref_n + fx[tag + "_count"]
        mean_rel = abs(float(color[..., :3].mean()) - float(fx[tag + "_color"][..., :3].mean())) / float(fx[tag + "_color"][..., :3].mean())
        assert mean_rel <= 5e-4, mean_rel
    if "strict_16spp_work" in fx:  # IEEE build of the reference: work totals agree to 1e-3, depth histogram too
        assert (np.abs(work - fx["strict_16spp_work"]) <= 1e-3 * fx["strict_16spp_work"]).all(), (work, fx["strict_16spp_work"])
    rms = cases.rms_per_channel(total_c, total_n, ref_c, ref_n).max()
    floor = float(fx["noise_floor_rms_16spp"].max()) if "noise_floor_rms_16spp" in fx else 0.0
    rep = cases.diff_report(total_c, total_n, ref_c, ref_n)
    print(f"{case}: oracle vs reference 16 spp rms {rms:.3e} (reference vs itself {floor:.3e}), {rep}, 1-spp {agree}")
    # either inside the reference's own build-to-build distance, or: a handful of pixels where one of the 16
    # samples took another branch, and rounding-level agreement everywhere else
    calibrated = rms <= max(2e-4, 3 * floor)
    few_flips = rep["n_flipped"] <= max(3, 0.005 * rep["n_pixels"]) and rep["rms_without_flipped"] <= 1e-4
    assert calibrated or few_flips, (rms, floor, rep)


@pytest.mark.parametrize("case", list(cases.CASES))
def test_oracle_equals_the_reference_strict_build(case, scene_factory):
    """Bit for bit: image, sample counts and histograms of iterations 0..3 as the reference's STRICT build produced them on
    the MI355X (fixtures `strict_exact_*`).  The oracle's arithmetic is that build's: IEEE operations in the written order
    plus the platform library's dot / cross / normalize (v_rsq_f32 through the measured table) / sin / cos."""
    fx = np.load(os.path.join(GOLDEN, f"ref_{case}.npz"))
    if "strict_exact_color" not in fx:
        pytest.skip("fixture predates the strict-build images")
    name, sampler, w, h, d = cases.CASES[case]
    if name == "tris1m" and os.environ.get("PTMI_SKIP_SLOW"):
        pytest.skip("slow")
    sc = scene_factory(name, w, h)
    color, count, (dep, bbx, tri), _ = O.oracle_render(sc, w, h, d, cases.STRICT_SPP, sampler=sampler)
    assert np.array_equal(count, fx["strict_exact_count"]) and np.array_equal(dep, fx["strict_exact_depths"])
    assert np.array_equal(bbx.astype(np.int64), _expand(fx["strict_exact_bbx_idx"], fx["strict_exact_bbx_val"]))
    assert np.array_equal(tri.astype(np.int64), _expand(fx["strict_exact_tri_idx"], fx["strict_exact_tri_val"]))
    bad = np.argwhere(color.view(np.uint32) != fx["strict_exact_color"].view(np.uint32))
    assert len(bad) == 0, f"{len(bad)} channel values differ from the reference's strict build, first at {bad[:5].tolist()}"


@pytest.mark.parametrize("case", list(cases.CASES))
def test_oracle_default_build_equals_the_reference_default_build(case, scene_factory):
    """Bit for bit: the oracle's DEFAULT-ARITHMETIC build (oracle/build/libpt_oracle_da.so: fused a*b+c where the OpenCL front
    end contracts, division through v_rcp_f32 and square roots through v_sqrt_f32 from the tables measured on the MI355X)
    against the committed outputs of the reference kernel as its own build line compiles it (fixtures `it0_8_*`, `it8_8_*`,
    `it0_1_color`: images, sample counts and the three histograms of two 8-iteration shards and of iteration 0 alone)."""
    fx = np.load(os.path.join(GOLDEN, f"ref_{case}.npz"))
    name, sampler, w, h, d = cases.CASES[case]
    if name == "tris1m" and os.environ.get("PTMI_SKIP_SLOW"):
        pytest.skip("slow")
    sc = scene_factory(name, w, h)
    one, _, _, _ = O.oracle_render(sc, w, h, d, 1, sampler=sampler, default_arithmetic=True)
    assert np.array_equal(one.view(np.uint32), fx["it0_1_color"].view(np.uint32))
    for first, n in cases.FIXTURE_RANGES:
        tag = f"it{first}_{n}"
        color, count, (dep, bbx, tri), _ = O.oracle_render(sc, w, h, d, n, first_iteration=first, sampler=sampler, default_arithmetic=True)
        assert np.array_equal(count, fx[tag + "_count"]) and np.array_equal(dep, fx[tag + "_depths"])
        assert np.array_equal(bbx.astype(np.int64), _expand(fx[tag + "_bbx_idx"], fx[tag + "_bbx_val"]))
        assert np.array_equal(tri.astype(np.int64), _expand(fx[tag + "_tri_idx"], fx[tag + "_tri_val"]))
        bad = np.argwhere(color.view(np.uint32) != fx[tag + "_color"].view(np.uint32))
        assert len(bad) == 0, f"{tag}: {len(bad)} channel values differ from the reference's default build, first at {bad[:5].tolist()}"


def test_oracle_default_build_equals_the_reference_default_build_feature_by_feature(built):
    """... and every material branch, light type, texture path and the cube-map sky in that arithmetic (`<feature>_default`
    digests of tests/golden/ref_feat_64x64_d8_features.npz)."""
    from opencl_pathtracer_amd import scenes, bvh_create
    fcase, w, h, d = cases.FEATURE_CASE
    fx = np.load(os.path.join(GOLDEN, f"ref_{fcase}_features.npz"))
    if scenes.FEATURES[0] + "_default" not in fx:
        pytest.skip("feature fixture predates the default-build digests")
    for feature in scenes.FEATURES:
        sc = bvh_create(scenes.build("feat_" + feature, w, h))
        color, count, (dep, bbx, tri), _ = O.oracle_render(sc, w, h, d, cases.FEATURE_SPP, default_arithmetic=True)
        assert cases.result_digest(color, count, dep, bbx, tri) == str(fx[feature + "_default"]), feature


def test_oracle_equals_the_reference_strict_build_feature_by_feature(built):
    """Every material branch, light type, texture path and the cube-map sky, one per scene: SHA-256 of the oracle's result
    equals the digest of the reference's strict build (tests/golden/ref_feat_64x64_d8_features.npz)."""
    from opencl_pathtracer_amd import scenes, bvh_create
    import importlib.util
    fcase, w, h, d = cases.FEATURE_CASE
    path = os.path.join(GOLDEN, f"ref_{fcase}_features.npz")
    if not os.path.exists(path):
        pytest.skip("feature fixture not generated yet")
    fx = np.load(path)
    spec = importlib.util.spec_from_file_location("mkfx", os.path.join(GOLDEN, "make_reference_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    for feature in scenes.FEATURES:
        sc = bvh_create(scenes.build("feat_" + feature, w, h))
        assert str(fx[feature + "_scene"]) == mk.scene_digest(sc), "scene generator changed since the fixture was made"
        color, count, (dep, bbx, tri), _ = O.oracle_render(sc, w, h, d, cases.FEATURE_SPP)
        assert cases.result_digest(color, count, dep, bbx, tri) == str(fx[feature]), feature


def test_oracle_equals_both_reference_builds_on_fuzzed_scenes(built):
    """Seeded scenes nobody arranged (scenes.fuzz_scene / corrupt_records: soup, ties, unsplittable leaves, all material and
    light types, records no importer writes, zero-area triangles whose NaN distances the reference accepts): the oracle in
    either arithmetic equals the digest of that build of the reference (tests/golden/ref_fuzz.npz, made on the MI355X)."""
    import importlib.util
    import warnings
    from opencl_pathtracer_amd import scenes, bvh_create
    path = os.path.join(GOLDEN, "ref_fuzz.npz")
    if not os.path.exists(path):
        pytest.skip("fuzz fixture not generated yet")
    fx = np.load(path)
    spec = importlib.util.spec_from_file_location("mkfx", os.path.join(GOLDEN, "make_reference_fixtures.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    for name in cases.FUZZ_FIXTURES:
        _, w, h, d = cases.FUZZ_CASE[int(name.rsplit("_l", 1)[1])]
        sc = cases.build_fuzz(name, w, h)
        assert str(fx[name + "_scene"]) == mk.full_scene_digest(sc), f"{name}: scene generator changed since the fixture was made"
        for da, key in ((False, name), (True, name + "_default")):
            color, count, (dep, bbx, tri), _ = O.oracle_render(sc, w, h, d, cases.FEATURE_SPP, default_arithmetic=da)
            assert cases.result_digest(color, count, dep, bbx, tri) == str(fx[key]), key


def test_oracle_under_sanitizers(tmp_path):
    """oracle/pt_oracle.c, both arithmetics, compiled with gcc -fsanitize=address,undefined,float-cast-overflow and run on the
    committed fuzz scenes: the checker itself reads nothing outside the scene's arrays and converts no NaN to an integer the way
    C leaves undefined (it restates the GPU's conversions: v_cvt gives 0 for NaN and saturates)."""
    import shutil
    import subprocess
    import sys
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("gcc not installed")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([gcc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip("libasan / libubsan not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = []
    for da in (0, 1):
        so = tmp_path / f"libpt_oracle_{da}_asan.so"
        r = subprocess.run([gcc, "-std=c11", "-O1", "-g", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fexcess-precision=standard", "-mfma",
                            "-pthread", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer",
                            f"-DPTO_DEFAULT_ARITHMETIC={da}", "-I" + os.path.join(root, "include"), "-shared",
                            os.path.join(root, "oracle", "pt_oracle.c"), "-o", str(so), "-lm"], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        libs.append(str(so))
    env = {**os.environ, "LD_PRELOAD": asan + ":" + ubsan, "ASAN_OPTIONS": "detect_leaks=0"}
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "sanitize", "oracle_asan.py"), *libs], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "clean" in r.stdout, (r.stdout[-500:], r.stderr[-3000:])
