"""Multi-GPU inside libptmi (SURVEY 8e behind the boundary): the iteration ids of every ptmi_render call are dealt to the
devices of the context modulo the device count, each device renders the full image for its ids on its own scene
replica, and ptmi_read_image / ptmi_read_snapshot sum the partial accumulators on devices[0] in device order.

CPU part: the id arithmetic (ptmi_device_share).  GPU part: on a one-GPU box the same device listed two or three
times runs the whole multi-device control flow (replicas, per-device streams, snapshots, peer copies, the sum kernel);
its image must equal the single-context one up to the order of the float additions, its counts, histograms and
counters exactly.  The snapshot ring (launches of the next images queued while image k is read back) is checked
against the blocking loop image by image.
"""
import numpy as np
import pytest

import cases
import oracle_ffi as O
from opencl_pathtracer_amd import Backend, PtmiError, render_scene, structs as S
from opencl_pathtracer_amd.backend import device_share
from opencl_pathtracer_amd import backend


@pytest.mark.parametrize("first,n,G", [(0, 16, 1), (0, 16, 2), (5, 1, 4), (7, 13, 3), (0, 4096, 8), (3, 2, 8), (10, 0, 4),
                                        (4294967000, 200, 7)])
def test_device_shares_partition_the_ids(first, n, G, built):
    ids = []
    for k in range(G):
        f, m = device_share(first, n, k, G)
        mine = [f + j * G for j in range(m)]
        assert all(i % G == k and first <= i < first + n for i in mine)
        ids += mine
    assert sorted(ids) == list(range(first, first + n))
    sizes = [device_share(first, n, k, G)[1] for k in range(G)]
    assert max(sizes) - min(sizes) <= 1  # balanced to one id


def test_device_share_single_calls_rotate(built):
    """The reference's loop renders one image per call (OpenCL.cpp:76-89): call i must land on device i mod G."""
    for G in (2, 3, 8):
        for i in range(20):
            owners = [k for k in range(G) if device_share(i, 1, k, G)[1] == 1]
            assert owners == [i % G]


@pytest.mark.gpu
@pytest.mark.parametrize("kernel_flags", [0, 2])
@pytest.mark.parametrize("G", [2, 3])
def test_same_device_listed_several_times_equals_single_context(G, kernel_flags, scene_factory):
    name, sampler, w, h, d = cases.CASES["matmix_96x96_d8"]
    sc = scene_factory(name, w, h)
    spp = 11  # not a multiple of G: shares of different sizes
    one, one_n, (dep, bbx, tri), counters = render_scene(sc, w, h, d, spp, flags=kernel_flags)
    m, m_n, (m_dep, m_bbx, m_tri), m_counters = render_scene(sc, w, h, d, spp, flags=kernel_flags, devices=[0] * G)
    assert np.array_equal(m_n, one_n) and m_counters == counters
    assert np.array_equal(m_dep, dep) and np.array_equal(m_bbx, bbx) and np.array_equal(m_tri, tri)
    assert (cases.rms_per_channel(m, m_n, one, one_n) <= 1e-6).all()
    assert np.allclose(m, one, rtol=2e-6, atol=1e-6)
    # and the parts are what the id arithmetic says: device k's share rendered alone, summed in device order, bit for bit
    parts = []
    for k in range(G):
        f, n = device_share(0, spp, k, G)
        be = Backend().setup_context(w, h, d, sc.lightsSize, sampler, flags=kernel_flags)
        be.initialize_memory(sc)
        for j in range(n):  # ids f, f + G, ...: one call each on a single-device context
            be.render(f + j * G, 1)
        parts.append(be.read_image())
        be.release()
    total = parts[0][0].copy()
    for c, _ in parts[1:]:
        total = total + c
    assert np.array_equal(total.view(np.uint32), m.view(np.uint32))


@pytest.mark.gpu
def test_one_image_per_call_over_two_devices(scene_factory):
    """The reference's call pattern (render(i, 1) per image) over a two-device context, readbacks in between."""
    w, h, d = 64, 48, 4
    sc = scene_factory("cornell", w, h)
    single = Backend().setup_context(w, h, d, sc.lightsSize)
    single.initialize_memory(sc)
    multi = Backend().setup_context(w, h, d, sc.lightsSize, devices=[0, 0])
    multi.initialize_memory(sc)
    for i in range(5):
        single.render(i, 1)
        multi.render(i, 1)
        a, an = single.read_image()
        b, bn = multi.read_image()
        assert np.array_equal(an, bn) and float(an.max()) == i + 1
        assert np.allclose(a, b, rtol=2e-6, atol=1e-6)
    assert single.counters() == multi.counters()
    bgr_s, bgr_m = single.read_display(), multi.read_display()
    assert np.abs(bgr_s.astype(int) - bgr_m.astype(int)).max() <= 1  # quantised after two summation orders
    with pytest.raises(PtmiError):
        multi.device_accumulators()  # partial sums are not handed out
    with pytest.raises(PtmiError):
        multi.set_stream(0)
    single.release()
    multi.release()


@pytest.mark.gpu
def test_super_sampling_on_two_devices_merges_the_moments(scene_factory):
    """SUPER_SAMPLING with the ids dealt to two devices: each samples adaptively on its own accumulators; the image is the
    sum, the variance image the pairwise merge of the two devices' moments - equal to merging, with
    distributed.merge_moments, two single-device contexts that rendered the same ids."""
    import torch
    from opencl_pathtracer_amd.distributed import merge_moments
    w, h, d, n = 64, 48, 4, 18
    sc = scene_factory("cornell", w, h)
    multi = Backend().setup_context(w, h, d, sc.lightsSize, super_sampling=True, devices=[0, 0])
    multi.initialize_memory(sc)
    multi.render(0, n)
    mc, mn = multi.read_image()
    mv = multi.read_variance()
    with pytest.raises(PtmiError):
        multi.device_variance()
    multi.release()
    parts = []
    for k in range(2):
        be = Backend().setup_context(w, h, d, sc.lightsSize, super_sampling=True)
        be.initialize_memory(sc)
        for it in range(k, n, 2):
            be.render(it, 1)
        c, cn = be.read_image()
        parts.append((c, cn, be.read_variance()))
        be.release()
    (ca, na, va), (cb, nb, vb) = parts
    assert np.array_equal(mn, na + nb) and mn.min() < n  # adaptive: some pixels were skipped
    assert np.array_equal(mc.view(np.uint32), (ca + cb).view(np.uint32))
    t = lambda a, shape: torch.from_numpy(np.ascontiguousarray(a).reshape(shape))
    _, _, m2 = merge_moments(t(ca, (-1, 4)), t(na, (-1,)), t(va, (-1, 4)), t(cb, (-1, 4)), t(nb, (-1,)), t(vb, (-1, 4)))
    assert np.isfinite(mv).all()
    assert np.allclose(mv.reshape(-1, 4), m2.numpy(), rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0]])
def test_snapshot_ring_equals_blocking_loop(devices, scene_factory):
    """Launches of images k+1, k+2 are queued BEFORE image k is read back: every snapshot still holds exactly images 0..k."""
    w, h, d = 96, 64, 6
    sc = scene_factory("tris20k", w, h)
    n_img, lookahead = 7, 2
    slots = lookahead + 1
    blocking = Backend().setup_context(w, h, d, sc.lightsSize, devices=devices)
    blocking.initialize_memory(sc)
    expect = []
    for i in range(n_img):
        blocking.render(i, 1)
        blocking.synchronize()
        c, n = blocking.read_image()
        expect.append((c.copy(), n.copy()))
    blocking.release()

    be = Backend().setup_context(w, h, d, sc.lightsSize, devices=devices)
    be.initialize_memory(sc)
    out = (np.empty((h, w, 4), np.float32), np.empty((h, w), np.float32))
    be.pin_host_buffer(out[0])  # colour goes by DMA, the count through the staging buffer
    queued = 0
    for i in range(n_img):
        while queued < n_img and queued <= i + lookahead:
            be.render(queued, 1)
            be.snapshot(queued % slots)
            queued += 1
        c, n = be.read_snapshot(i % slots, out=out)
        assert np.array_equal(n, expect[i][1]), i
        assert np.array_equal(c.view(np.uint32), expect[i][0].view(np.uint32)), i
    with pytest.raises(PtmiError):
        be.read_snapshot(slots + 1)  # never filled
    with pytest.raises(PtmiError):
        be.snapshot(backend.USER_SNAPSHOT_SLOTS)  # the library's own slot
    be.unpin_host_buffer(out[0])
    with pytest.raises(PtmiError):
        be.unpin_host_buffer(out[1])  # never pinned
    c, n = be.read_image(out=out)  # staging path again
    assert np.array_equal(c.view(np.uint32), expect[-1][0].view(np.uint32))
    be.release()


@pytest.mark.gpu
def test_eight_listed_devices_equal_the_single_context(scene_factory):
    """BASELINE configs[3]'s device count (one GPU listed eight times here): ids spread over eight shares, eight partial images
    summed on devices[0] - counts, histograms and counters equal the single-context render exactly, the image up to the order of
    the float additions."""
    w, h, d, n = 96, 64, 6, 24
    sc = scene_factory("tris20k", w, h)
    color, count, stats, counters = render_scene(sc, w, h, d, n)
    c8, n8, s8, k8 = render_scene(sc, w, h, d, n, devices=[0] * 8)
    assert np.array_equal(n8, count) and k8 == counters and all(np.array_equal(a, b) for a, b in zip(stats, s8))
    assert np.allclose(c8, color, rtol=2e-6, atol=1e-6)
    # ... and equals the sum of its eight parts rendered alone, in device order, bit for bit
    acc = np.zeros_like(color)
    for k in range(8):
        be = Backend().setup_context(w, h, d, sc.lightsSize)
        be.initialize_memory(sc)
        for j in range(n // 8):
            be.render(k + 8 * j, 1)
        part, _ = be.read_image()
        be.release()
        acc = acc + part if k else part.copy()
    assert np.array_equal(c8.view(np.uint32), acc.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("scene,w,h,d", [("tris1m", 1920, 1080, 10), ("mayalike", 3840, 2160, 16)], ids=["configs3", "configs4"])
def test_eight_listed_devices_on_the_baseline_workloads(scene, w, h, d):
    """BASELINE configs[3] and configs[4] are eight-GPU jobs: their SCENES at their SIZES through the eight-device control flow
    (one GPU listed eight times: eight scene replicas, eight stage sets, eight partial images of 41 / 166 MB summed on
    devices[0] in device order), 16 iterations = two per device, in the reference's own arithmetic: counts, histograms and
    counters equal the single-context render exactly, the image up to the order of the float additions."""
    from opencl_pathtracer_amd import scenes, bvh_create
    sc = bvh_create(scenes.build(scene, w, h))
    flags = backend.FLAG_DEFAULT_ARITHMETIC
    color, count, stats, counters = render_scene(sc, w, h, d, 16, flags=flags)
    c8, n8, s8, k8 = render_scene(sc, w, h, d, 16, flags=flags, devices=[0] * 8)
    assert np.array_equal(n8, count) and k8 == counters and all(np.array_equal(a, b) for a, b in zip(stats, s8))
    assert np.allclose(c8, color, rtol=4e-6, atol=1e-6)
    assert (cases.rms_per_channel(c8, n8, color, count) <= 1e-6).all()


@pytest.mark.gpu
def test_rccl_reduce_behind_the_c_abi(scene_factory, monkeypatch):
    """The collective north_star names, reached through the reference API's readback: PTMI_REDUCE=rccl-always sends the image of
    a (one-device) context through librccl's ncclReduce - loaded at run time, one-rank communicator from ncclCommInitAll, on the
    context's copy stream - before it crosses the bus.  (With several DISTINCT devices PTMI_REDUCE=rccl opts into that path,
    peer copies in device order being the default; a one-GPU box can only check the plumbing and that the sum of one share is
    that share.)"""
    w, h, d, n = 64, 48, 4, 5
    sc = scene_factory("cornell", w, h)
    color, count, _, _ = render_scene(sc, w, h, d, n)
    monkeypatch.setenv("PTMI_REDUCE", "rccl-always")
    be = Backend().setup_context(w, h, d, sc.lightsSize)
    be.initialize_memory(sc)
    be.render(0, n)
    be.snapshot(3)
    c, cn = be.read_snapshot(3)
    be.release()
    assert np.array_equal(c.view(np.uint32), color.view(np.uint32)) and np.array_equal(cn, count)
    # devices listed twice are refused by RCCL: the peer-copy sum takes over, silently
    c2, n2, _, _ = render_scene(sc, w, h, d, n, devices=[0, 0])
    assert np.array_equal(n2, count) and np.allclose(c2, color, rtol=2e-6, atol=1e-6)


@pytest.mark.gpu
def test_multi_device_bit_exact_parts_vs_oracle(scene_factory):
    """Each device's share is the oracle's image of exactly those iteration ids (the samples are a pure function of
    pixel and id, header.cl:255-264): two devices, ids 0,2,4 and 1,3,5."""
    w, h, d = 64, 48, 4
    sc = scene_factory("cornell", w, h)
    parts = []
    for k in range(2):
        acc = np.zeros((h, w, 4), np.float32)
        for j in range(3):
            c, _, _, _ = O.oracle_render(sc, w, h, d, 1, first_iteration=k + 2 * j)
            acc = acc + c
        parts.append(acc)
    expect = parts[0] + parts[1]
    color, count, _, _ = render_scene(sc, w, h, d, 6, devices=[0, 0])
    assert float(count.min()) == 6.0 == float(count.max())
    assert np.array_equal(color.view(np.uint32), expect.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0], [0] * 8])
def test_render_snapshots_leaves_every_image_of_the_per_image_loop(devices, scene_factory):
    """ptmi_render_snapshots(first, n): iterations share one launch, yet slot k holds exactly
    what the reference's loop would have read back after image first + k.  With G devices a device copies its accumulators only
    when they changed (every G-th image) and devices[0] receives only the share that changed: same images."""
    w, h, d = 96, 64, 6
    sc = scene_factory("tris20k", w, h)
    first, n = 3, 20
    blocking = Backend().setup_context(w, h, d, sc.lightsSize, devices=devices)
    blocking.initialize_memory(sc)
    blocking.render(0, first)
    expect = []
    for i in range(first, first + n):
        blocking.render(i, 1)
        c, cn = blocking.read_image()
        expect.append((c.copy(), cn.copy()))
    stats = blocking.read_statistics()
    counters = blocking.counters()
    blocking.release()

    be = Backend().setup_context(w, h, d, sc.lightsSize, devices=devices)
    be.initialize_memory(sc)
    be.render(0, first)
    ring = backend.USER_SNAPSHOT_SLOTS
    be.render_snapshots(first, n, first_slot=ring - 2)  # wraps around the caller slots
    for k in range(n):
        c, cn = be.read_snapshot((ring - 2 + k) % ring)
        assert np.array_equal(cn, expect[k][1]), k
        assert np.array_equal(c.view(np.uint32), expect[k][0].view(np.uint32)), k
    s2 = be.read_statistics()
    assert all(np.array_equal(a, b) for a, b in zip(stats, s2)) and be.counters() == counters
    final, _ = be.read_image()
    assert np.array_equal(final.view(np.uint32), expect[-1][0].view(np.uint32))
    with pytest.raises(PtmiError):
        be.render_snapshots(0, ring + 1)
    # a new scene on the same context: the ring is empty again (a slot of the old scene must not be readable)
    be.initialize_memory(sc)
    with pytest.raises(PtmiError) as e:
        be.read_snapshot(0)
    assert e.value.code == -6
    be.render_snapshots(0, 2)
    c, cn = be.read_snapshot(1)
    assert float(cn.min()) == 2.0 == float(cn.max())
    be.release()
    rnd = Backend().setup_context(w, h, d, sc.lightsSize, sampler=S.RANDOM)
    rnd.initialize_memory(sc)
    with pytest.raises(PtmiError) as e:
        rnd.render_snapshots(0, 2)
    assert e.value.code == -7
    rnd.release()
