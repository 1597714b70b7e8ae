"""Adaptive super-sampling (-D SUPER_SAMPLING): the chi-square table, the oracle's behaviour, and on the GPU the
HIP path against the oracle (bit-exact) and against the reference kernel built with the same define."""
import os
import re

import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import Backend, PtmiError, render_scene, structs as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TABLE = "/root/reference/Kernel/X2inv.cl"


def product_table():
    text = open(os.path.join(ROOT, "opencl_pathtracer_amd", "csrc", "x2inv_table.inc")).read()
    text = re.sub(r"//.*", "", text)
    return np.array([float(v.strip().rstrip("f")) for v in text.split(",") if v.strip()], np.float32)


def test_generated_table():
    t = product_table()
    assert len(t) == 1001 and t[0] == 0 and np.all(np.diff(t) > 0)
    assert np.array_equal(t, O.x2inv_table())  # product's generated file == definition evaluated in the test
    assert abs(t[1000] - 898.912447) < 1e-3 and abs(t[1] - 0.000157) < 1e-6


@pytest.mark.skipif(not os.path.exists(REF_TABLE), reason="reference tree not present")
def test_table_equals_the_reference_file():
    txt = open(REF_TABLE).read()
    body = txt[txt.index("{") + 1:txt.rindex("}")]
    ref = np.array([float(v.strip().rstrip("f")) for v in body.split(",") if v.strip()], np.float32)
    assert np.array_equal(ref, product_table())


def test_oracle_skips_converged_pixels(scene_factory):
    """cl:1219-1222: from iteration 6 on a pixel is sampled with probability ~ its relative variance + 5 %."""
    sc = scene_factory("cornell", 64, 48)
    color, count, (dep, _, _), tot = O.oracle_render(sc, 64, 48, 4, 24, super_sampling=True)
    assert count.max() == 24 and count.min() >= 6 and count.min() < 24  # first 6 iterations never skip
    assert dep.sum() == tot["paths"] == int(count.sum()) < 64 * 48 * 24
    plain, plain_n, _, _ = O.oracle_render(sc, 64, 48, 4, 24)
    # same estimator: the adaptive image stays close to the plain one where it kept sampling
    a = color[..., :3] / count[..., None]
    b = plain[..., :3] / plain_n[..., None]
    assert np.abs(a - b).mean() < 0.05
    # deterministic
    again = O.oracle_render(sc, 64, 48, 4, 24, super_sampling=True)
    assert np.array_equal(again[0], color) and np.array_equal(again[1], count)


def test_unsupported_combinations_fail_loudly(built):
    from opencl_pathtracer_amd import backend
    lib = backend.load_library()
    if lib.ptmi_device_count() == 0:
        pytest.skip("argument check happens after device discovery only for valid configs; needs no GPU otherwise")
    with pytest.raises(PtmiError) as e:
        backend.Backend().setup_context(8, 8, 2, 0, super_sampling=True, flags=backend.FLAG_MEGAKERNEL)
    assert e.value.code == -7


@pytest.mark.gpu
def test_super_sampling_with_the_random_sampler(scene_factory):
    """SUPER_SAMPLING x SAMPLE_RANDOM, a combination the reference compiles and runs (FullKernel.cl:1137-1141 with :1152-1172,
    :1219-1222): samples land on arbitrary pixels, so the stop test reads - and the epilogue updates - the count and variance
    of a pixel other work-items are updating too.  The reference races there; here every update is an atomic.  Against the
    serial oracle the comparison is therefore statistical: the same estimator and nearly the same stop decisions."""
    w, h, d, n = 64, 48, 4, 24
    sc = scene_factory("cornell", w, h)
    be = Backend().setup_context(w, h, d, sc.lightsSize, sampler=S.RANDOM, super_sampling=True)
    be.initialize_memory(sc)
    be.render(0, n)
    color, count = be.read_image()
    var = be.read_variance()
    (dep, _, _), counters = be.read_statistics(), be.counters()
    be.release()
    o_color, o_count, (o_dep, _, _), totals = O.oracle_render(sc, w, h, d, n, sampler=S.RANDOM, super_sampling=True)
    assert np.isfinite(color).all()
    assert int(count.sum()) == counters["paths"] == int(dep.sum()) < w * h * n  # some paths were skipped ...
    assert abs(int(count.sum()) - int(o_count.sum())) <= 0.02 * o_count.sum()  # ... about as many as the serial evaluation skips
    assert np.abs(dep.astype(np.int64) - o_dep.astype(np.int64)).sum() <= 0.03 * o_dep.sum()
    a = color[..., :3].sum(axis=(0, 1)) / count.sum()
    b = o_color[..., :3].sum(axis=(0, 1)) / o_count.sum()
    assert np.allclose(a, b, rtol=0.02)
    # a pixel whose first sample came after iteration 0 has a NaN variance in the reference (0 / 0, :1349) and is never skipped
    o_var = np.zeros((h, w, 4), np.float32)
    O.oracle_render(sc, w, h, d, n, sampler=S.RANDOM, super_sampling=True, image_v=o_var)
    assert abs(int(np.isnan(var[..., 0]).sum()) - int(np.isnan(o_var[..., 0]).sum())) <= 0.05 * np.isnan(o_var[..., 0]).sum() + 5
    ok = ~np.isnan(var[..., :3])
    assert (var[..., :3][ok] >= -1e-3).all() and var[..., :3][ok].max() > 0  # sums of products of same-signed deviations


@pytest.mark.gpu
@pytest.mark.parametrize("name,w,h,d,sampler", [("cornell", 64, 48, 4, S.JITTERED), ("matmix", 96, 96, 8, S.UNIFORM)])
def test_super_sampling_bit_exact_vs_oracle(name, w, h, d, sampler, scene_factory):
    sc = scene_factory(name, w, h)
    n = 20
    color, count, (dep, bbx, tri), counters = render_scene(sc, w, h, d, n, sampler=sampler, super_sampling=True)
    o_color, o_count, (o_dep, o_bbx, o_tri), totals = O.oracle_render(sc, w, h, d, n, sampler=sampler, super_sampling=True)
    assert np.array_equal(count, o_count) and count.min() < n
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32))
    assert np.array_equal(dep, o_dep) and np.array_equal(bbx, o_bbx) and np.array_equal(tri, o_tri) and counters == totals


@pytest.mark.gpu
def test_super_sampling_bit_exact_vs_reference_default_build(scene_factory):
    """The reference kernel built -D SUPER_SAMPLING as its own build line builds it, 32 iterations: with the JITTERED sampler every
    work-item owns its pixel, so the adaptive render is deterministic - and in the default-arithmetic mode the integrator makes
    the same stop decisions, sample for sample: image, sample counts (the sampling density map) and depth histogram are EQUAL."""
    from opencl_pathtracer_amd import backend
    case = "cornell_64x48_d4_ss"
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    sc = scene_factory("cornell", 64, 48)
    n = 32
    r_color, r_count, (r_dep, r_bbx, r_tri), _ = O.ref_gpu_render(case, sc, 64, 48, 4, n)
    color, count, (dep, bbx, tri), _ = render_scene(sc, 64, 48, 4, n, super_sampling=True, flags=backend.FLAG_DEFAULT_ARITHMETIC)
    assert count.min() < n  # (pixels were skipped)
    assert np.array_equal(count, r_count) and np.array_equal(dep, r_dep) and np.array_equal(bbx, r_bbx) and np.array_equal(tri, r_tri)
    assert np.array_equal(color.view(np.uint32), r_color.view(np.uint32))
    o_color, o_count, _, _ = O.oracle_render(sc, 64, 48, 4, n, super_sampling=True, default_arithmetic=True)
    assert np.array_equal(o_count, r_count) and np.array_equal(o_color.view(np.uint32), r_color.view(np.uint32))


@pytest.mark.gpu
def test_super_sampling_vs_reference_kernel(scene_factory):
    """The reference kernel built with -D SUPER_SAMPLING: same sampling density map (statistically) and image."""
    case = "cornell_64x48_d4_ss"
    if not O.have_ref_kernel(case):
        O.missing_reference("oracle/_ref code object not present")
    sc = scene_factory("cornell", 64, 48)
    n = 32
    r_color, r_count, (r_dep, _, _), _ = O.ref_gpu_render(case, sc, 64, 48, 4, n)
    color, count, (dep, _, _), _ = render_scene(sc, 64, 48, 4, n, super_sampling=True)
    assert count.max() == r_count.max() == n and (count[..., None] >= 6).all()
    # decisions compare a random number with a variance estimate: they flip where arithmetic differs in the last
    # bits, so totals agree statistically, not sample by sample
    assert abs(float(count.sum()) - float(r_count.sum())) <= 0.02 * float(r_count.sum())
    assert abs(int(dep.sum()) - int(r_dep.sum())) <= 0.02 * int(r_dep.sum())
    a = color[..., :3] / count[..., None]
    b = r_color[..., :3] / r_count[..., None]
    print("ss vs reference: samples", int(count.sum()), int(r_count.sum()), "mean abs image diff", float(np.abs(a - b).mean()))
    assert np.abs(a - b).mean() < 0.02


@pytest.mark.gpu
def test_variance_image_readback_and_two_shard_merge(scene_factory):
    """ptmi_read_variance returns the kernel's imageV bit for bit (vs the oracle), fails without SUPER_SAMPLING, and two
    shards of iteration ids merge (distributed.merge_moments) into the moments of all their samples."""
    import torch
    from opencl_pathtracer_amd import Backend
    from opencl_pathtracer_amd.distributed import merge_moments
    w, h, d = 64, 48, 4
    sc = scene_factory("cornell", w, h)
    o_v = np.zeros((h, w, 4), np.float32)
    O.oracle_render(sc, w, h, d, 12, super_sampling=True, image_v=o_v)
    be = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
    be.initialize_memory(sc)
    be.render(0, 12)
    v = be.read_variance()
    assert np.array_equal(v.view(np.uint32), o_v.view(np.uint32)) and be.device_variance()
    be.release()
    plain = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED)
    plain.initialize_memory(sc)
    with pytest.raises(PtmiError):
        plain.read_variance()
    plain.release()
    # two shards of 3 iterations each, all below the first adaptive iteration (it > 5, cl:1219) so that every pixel takes
    # every sample: ids 0..2 on one context, 3..5 on another; merged moments == one context over 0..5 up to fp32 rounding
    shards = []
    for first in (0, 3):
        b = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
        b.initialize_memory(sc)
        b.render(first, 3)
        c, n = b.read_image()
        shards.append((c, n, b.read_variance()))
        b.release()
    whole = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
    whole.initialize_memory(sc)
    whole.render(0, 6)
    wc, wn = whole.read_image()
    wv = whole.read_variance()
    whole.release()
    t = lambda a, shape: torch.from_numpy(a.reshape(shape))
    (ca, na, va), (cb, nb, vb) = shards
    # the second shard starts at iteration 3 != 0: its first sample per pixel is treated like iteration 0 (no deviation
    # yet) instead of the reference's 0/0, so its variance image is finite and is the real M2 of its three samples
    assert np.isfinite(vb).all() and np.isfinite(va).all()
    s, n, m2 = merge_moments(t(ca, (-1, 4)), t(na, (-1,)), t(va, (-1, 4)), t(cb, (-1, 4)), t(nb, (-1,)), t(vb, (-1, 4)))
    assert np.array_equal(n.numpy().reshape(h, w), wn)
    assert np.allclose(s.numpy().reshape(h, w, 4), wc, rtol=1e-5, atol=1e-6)
    # merged M2 == the single context's variance image (two different fp32 summation orders of the same six samples)
    m2 = m2.numpy().reshape(h, w, 4)
    scale = np.maximum(np.abs(wv), 1e-3 * np.abs(wv).max())
    assert (np.abs(m2 - wv) / scale).max() < 1e-3, float((np.abs(m2 - wv) / scale).max())


@pytest.mark.gpu
def test_resume_with_variance(scene_factory):
    """ptmi_write_image + ptmi_write_variance restore a saved adaptive render: continuing it equals never stopping."""
    from opencl_pathtracer_amd import Backend
    w, h, d = 64, 48, 4
    sc = scene_factory("cornell", w, h)
    whole = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
    whole.initialize_memory(sc)
    whole.render(0, 16)
    wc, wn = whole.read_image()
    wv = whole.read_variance()
    whole.release()
    a = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
    a.initialize_memory(sc)
    a.render(0, 9)
    c, n = a.read_image()
    v = a.read_variance()
    a.release()
    b = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, super_sampling=True)
    b.initialize_memory(sc)
    b.write_image(c, n)
    b.write_variance(v)
    b.render(9, 7)
    rc, rn = b.read_image()
    rv = b.read_variance()
    b.release()
    assert np.array_equal(rn, wn) and rn.min() < 16
    assert np.array_equal(rc.view(np.uint32), wc.view(np.uint32)) and np.array_equal(rv.view(np.uint32), wv.view(np.uint32))
