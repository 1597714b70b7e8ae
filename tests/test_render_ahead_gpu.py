"""Rendering ahead of a blocking caller (round 4).  The reference's loop asks for ONE image per call and waits for it
(OpenCL.cpp:76-107: launch, clFinish, read, callback); a caller of the C ABI that does the same would leave the GPU idle while
it waits, reads and shows, and every launch would pay its ramp-up and its ragged end alone.  So once a context has SEEN a caller
come back for the next ids (same count), ptmi_render keeps the launches of the next calls in flight before they are asked for
(PTMI_RENDER_AHEAD, default 2), each on a stage set and with a counter block of its own, and ADOPTS one when its call comes -
staged radiances into the accumulators, statistics words into the histograms, counters into the totals - or drops it without a
trace when the caller asks for something else.  ONE such launch renders for up to four calls (PTMI_RENDER_AHEAD_CALLS, growing
with the calls the caller has come back in a row): each call adopts its part of the staging arrays and its own block of the
launch's counters.  Invisible by construction; this file checks that it is."""
import numpy as np
import pytest

import cases
from opencl_pathtracer_amd import Backend, structs as S
from opencl_pathtracer_amd import backend

pytestmark = pytest.mark.gpu
DA = backend.FLAG_DEFAULT_ARITHMETIC


def _play(sc, w, h, d, calls, flags, monkeypatch, ahead, sampler=S.JITTERED, calls_per_launch=None, devices=None):
    """calls: ("render", first, n) | ("clear",) | ("read",) | ("counters",) -> list of what the reads returned + the final state"""
    monkeypatch.setenv("PTMI_RENDER_AHEAD", str(ahead))
    if calls_per_launch is None:
        monkeypatch.delenv("PTMI_RENDER_AHEAD_CALLS", raising=False)
    else:
        monkeypatch.setenv("PTMI_RENDER_AHEAD_CALLS", str(calls_per_launch))
    be = Backend().setup_context(w, h, d, sc.lightsSize, sampler, flags=flags, devices=devices)
    seen = []
    try:
        be.initialize_memory(sc)
        for c in calls:
            if c[0] == "render":
                be.render(c[1], c[2])
                be.synchronize()  # the blocking caller
            elif c[0] == "clear":
                be.clear()
            elif c[0] == "read":
                color, count = be.read_image()
                seen.append((color.copy(), count.copy()))
            elif c[0] == "counters":
                seen.append(be.counters())
        color, count = be.read_image()
        stats = be.read_statistics()
        counters = be.counters()
        retraced = be.scheduler_stats()["paths_retraced"]
    finally:
        be.release()
    return seen, color, count, stats, counters, retraced


def _same(a, b):
    sa, ca, na, ta, ka, ra = a
    sb, cb, nb, tb, kb, rb = b
    assert np.array_equal(ca.view(np.uint32), cb.view(np.uint32)) and np.array_equal(na, nb)
    assert all(np.array_equal(x, y) for x, y in zip(ta, tb)) and ka == kb and ra == rb
    assert len(sa) == len(sb)
    for x, y in zip(sa, sb):
        if isinstance(x, dict):
            assert x == y
        else:
            assert np.array_equal(x[0].view(np.uint32), y[0].view(np.uint32)) and np.array_equal(x[1], y[1])


SEQUENCES = {
    # the reference's loop: one image per call, in order
    "one_by_one": [("render", k, 1) for k in range(9)],
    # ... with reads and counter reads in between (what OpenCL_RunKernel does after every image)
    "one_by_one_read_each": [x for k in range(7) for x in (("render", k, 1), ("read",), ("counters",))],
    # the caller leaves the pattern: jumps, repeats, other counts, a long call, a clear in the middle
    "jumps": [("render", 0, 1), ("render", 1, 1), ("render", 2, 1), ("render", 7, 1), ("render", 8, 1), ("render", 8, 1), ("counters",),
              ("render", 9, 2), ("render", 11, 2), ("render", 13, 2), ("render", 20, 3), ("render", 23, 3), ("render", 26, 1), ("read",),
              ("render", 27, 40), ("render", 67, 1), ("render", 68, 1), ("clear",), ("render", 0, 1), ("render", 1, 1), ("render", 2, 1),
              ("render", 1, 1), ("render", 2, 1), ("render", 3, 1), ("counters",)],
    # ... long enough for launches that render for four calls (the fourth launch ahead is the first of that size), the counters
    # read after every call, the image now and then; then pairs (two calls per launch), a jump into the middle of a launch that
    # ran ahead, and on
    "one_by_one_long": [x for k in range(22) for x in ((("render", k, 1), ("counters",)) + ((("read",),) if k % 5 == 4 else ()))] +
                       [x for k in range(6) for x in (("render", 22 + 2 * k, 2), ("counters",))] +
                       [("render", 40, 1), ("render", 41, 1), ("render", 42, 1), ("render", 43, 1), ("render", 44, 1), ("render", 45, 1),
                        ("render", 46, 1), ("render", 47, 1), ("render", 48, 1), ("render", 50, 1), ("counters",), ("render", 51, 1),
                        ("render", 52, 1), ("render", 53, 1), ("render", 54, 1), ("render", 55, 1), ("render", 56, 1), ("clear",),
                        ("render", 57, 1), ("render", 58, 1), ("counters",)],
    # pairs and triples in order
    "pairs_then_triples": [("render", 2 * k, 2) for k in range(5)] + [("render", 10 + 3 * k, 3) for k in range(4)],
}


@pytest.mark.parametrize("name", list(SEQUENCES))
@pytest.mark.parametrize("case", ["cornell_64x48_d4", "matmix_96x96_d8"])
def test_rendering_ahead_is_invisible(case, name, scene_factory, monkeypatch):
    scene, sampler, w, h, d = cases.CASES[case]
    sc = scene_factory(scene, w, h)
    plain = _play(sc, w, h, d, SEQUENCES[name], DA, monkeypatch, ahead=0)
    for depth in (1, 2, 3):
        _same(_play(sc, w, h, d, SEQUENCES[name], DA, monkeypatch, ahead=depth), plain)
    for calls_per_launch in (1, 2, 3):  # (above: the default, four)
        _same(_play(sc, w, h, d, SEQUENCES[name], DA, monkeypatch, ahead=2, calls_per_launch=calls_per_launch), plain)


def test_rendering_ahead_with_the_statistics_build_and_no_histograms(scene_factory, monkeypatch):
    scene, sampler, w, h, d = cases.CASES["tris20k_96x64_d6"]
    sc = scene_factory(scene, w, h)
    for flags in (DA | backend.FLAG_SCHEDULER_STATS, backend.FLAG_NO_HISTOGRAMS, 0):
        _same(_play(sc, w, h, d, SEQUENCES["one_by_one_read_each"], flags, monkeypatch, ahead=2),
              _play(sc, w, h, d, SEQUENCES["one_by_one_read_each"], flags, monkeypatch, ahead=0))


def test_rendering_ahead_of_paths_that_are_given_up(monkeypatch):
    """A launch that ran ahead hands its given-up paths to the literal loops on its own stage set, with its own counter block."""
    import warnings
    from opencl_pathtracer_amd import scenes, bvh_create
    w, h, d = 64, 64, 8
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        wild = bvh_create(scenes.build("fuzz3h_l1", w, h))
    b = _play(wild, w, h, d, SEQUENCES["one_by_one"], DA, monkeypatch, ahead=0)
    for calls_per_launch in (1, None):
        a = _play(wild, w, h, d, SEQUENCES["one_by_one"], DA, monkeypatch, ahead=2, calls_per_launch=calls_per_launch)
        assert a[5] > 0
        _same(a, b)
    # ... and in launches that render for four calls: every call gets the re-traced paths' counts of ITS iterations
    long_run = [x for k in range(20) for x in (("render", k, 1), ("counters",))]
    _same(_play(wild, w, h, d, long_run, DA, monkeypatch, ahead=2), _play(wild, w, h, d, long_run, DA, monkeypatch, ahead=0))


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["two", "three"])
def test_rendering_ahead_per_device_in_a_multi_device_context(devices, scene_factory, monkeypatch):
    """A context of G devices gives device k the ids = k (mod G): a caller that asks for one image per call comes to each device
    with every G-th call, and each device renders ahead of ITS calls (stride G) - without that only one of the G devices would
    work at a time.  Same device list, with and without: every read, the counters after every call and the final state equal bit
    for bit (the partial images are summed in device order either way)."""
    scene, sampler, w, h, d = cases.CASES["matmix_96x96_d8"]
    sc = scene_factory(scene, w, h)
    for name in ("one_by_one_long", "jumps"):
        _same(_play(sc, w, h, d, SEQUENCES[name], DA, monkeypatch, ahead=2, devices=devices),
              _play(sc, w, h, d, SEQUENCES[name], DA, monkeypatch, ahead=0, devices=devices))


def test_rendering_ahead_is_not_used_where_it_cannot_be(scene_factory, monkeypatch):
    """RANDOM sampler (nothing staged), SUPER_SAMPLING (every iteration reads the accumulators); and a device listed twice against
    one device: the same results as ever (the RANDOM sampler's float sums are atomic: counts and totals exactly, colours closely)."""
    scene, _, w, h, d = cases.CASES["cornell_64x48_d4"]
    sc = scene_factory(scene, w, h)
    calls = SEQUENCES["one_by_one"]
    a = _play(sc, w, h, d, calls, DA, monkeypatch, ahead=2, sampler=S.RANDOM)
    b = _play(sc, w, h, d, calls, DA, monkeypatch, ahead=0, sampler=S.RANDOM)
    assert np.array_equal(a[2], b[2]) and a[4] == b[4] and np.allclose(a[1], b[1], rtol=1e-4, atol=1e-5)
    monkeypatch.setenv("PTMI_RENDER_AHEAD", "2")
    outs = []
    for devices in (None, [0, 0]):
        be = Backend().setup_context(w, h, d, sc.lightsSize, S.JITTERED, flags=DA, devices=devices)
        try:
            be.initialize_memory(sc)
            for k in range(6):
                be.render(k, 1)
                be.synchronize()
            outs.append((be.read_image(), be.counters()))
        finally:
            be.release()
    assert np.array_equal(outs[0][0][1], outs[1][0][1]) and outs[0][1] == outs[1][1]
    assert np.allclose(outs[0][0][0], outs[1][0][0], rtol=2e-6, atol=1e-6)
