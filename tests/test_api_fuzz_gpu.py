"""Random sequences of C-ABI calls against a model of what they mean (GPU).

The calls of include/ptmi.h in random order - renders of arbitrary iteration ranges, runs of short blocking calls in order (which the library
renders ahead of), snapshots into arbitrary ring slots, bursts (ptmi_render_snapshots), reads of the image / a slot / the statistics / the counters, clears, re-uploads, image writes - on one
device and on one device listed two or three times (the in-library multi-device path with its lazy snapshots and incremental
peer copies).  The model: the accumulators are the sum of the per-iteration images rendered since the last clear / upload /
write, a slot holds the accumulators as they were when its snapshot was queued; the per-iteration images come from the CPU
oracle.  One device: bit for bit.  Several: sample counts and statistics exactly, sums up to their association."""
import os

import numpy as np
import pytest

import oracle_ffi as O
from opencl_pathtracer_amd import Backend, PtmiError, backend, structs as S

pytestmark = pytest.mark.gpu
W, H, D, N_IDS = 40, 30, 3, 48


SCENES = ["cornell", "fuzz3_l1", "fuzz5h_l1", "fuzz41hr_l1"]  # (one light each; the last two hold NaN-distance records: the NANSAFE instantiation)


@pytest.fixture(scope="module")
def per_iteration():
    """{(scene name, default arithmetic?): (scene, [per-iteration (color, count, statistics, totals) from the oracle])}"""
    import warnings
    from opencl_pathtracer_amd import scenes, bvh_create
    out = {}
    for name in SCENES:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sc = bvh_create(scenes.build(name, W, H))
        for da in (False, True):
            out[name, da] = (sc, [O.oracle_render(sc, W, H, D, 1, first_iteration=k, default_arithmetic=da) for k in range(N_IDS)])
    return out


class Model:
    def __init__(self, per_it):
        self.per_it = per_it
        self.reset()
        self.slots = {}

    def reset(self):
        self.color = np.zeros((H, W, 4), np.float32)
        self.count = np.zeros((H, W), np.float32)
        self.stats = [np.zeros(D + 1, np.int64), np.zeros(S.MAX_INTERSETCION_NUMBER, np.int64), np.zeros(S.MAX_INTERSETCION_NUMBER, np.int64)]
        self.totals = None

    def add(self, k):
        c, n, st, tot = self.per_it[k]
        self.color = self.color + c  # float32 adds, in call order: what a single device does
        self.count = self.count + n
        for a, b in zip(self.stats, st):
            a += b
        self.totals = dict(tot) if self.totals is None else {key: self.totals[key] + tot[key] for key in tot}


@pytest.mark.parametrize("devices", [None, [0, 0], [0, 0, 0]], ids=["one", "two", "three"])
@pytest.mark.parametrize("seed", range(int(os.environ.get("PTMI_API_FUZZ_SEEDS", "24"))))  # (a soak: PTMI_API_FUZZ_SEEDS=400)
def test_random_call_sequences(seed, devices, per_iteration):
    rs = np.random.RandomState(1234 + seed)
    exact = devices is None
    da = seed % 3 != 0
    flags = (backend.FLAG_DEFAULT_ARITHMETIC if da else 0) | (backend.FLAG_MEGAKERNEL if seed % 5 == 4 else 0)
    bursts = not flags & backend.FLAG_MEGAKERNEL  # (ptmi_render_snapshots wants the wavefront kernel - or a scene that forces the other)
    sc, per_it = per_iteration[SCENES[0], da]
    be = Backend().setup_context(W, H, D, sc.lightsSize, S.JITTERED, flags=flags, devices=devices)
    m = Model(per_it)
    pinned = [np.empty((H, W, 4), np.float32), np.empty((H, W), np.float32)]
    is_pinned = False
    ring = backend.USER_SNAPSHOT_SLOTS
    few = ring if seed % 2 else 6  # (even seeds: six slots, so that snapshots overwrite each other - and the slots lazy copies point to - all the time)
    log = []

    def same_image(got, want_color, want_count, what):
        color, count = got
        history = f"{what}; calls so far: ... " + " ".join(str(c) for c in log[-40:])
        if not np.array_equal(count, want_count):
            raise AssertionError(f"sample counts {np.unique(count).tolist()} instead of {np.unique(want_count).tolist()}: " + history)
        if exact and not np.array_equal(color.view(np.uint32), want_color.view(np.uint32)):
            raise AssertionError("image bits differ: " + history)
        if not exact and not np.allclose(color, want_color, rtol=3e-6, atol=1e-6):
            raise AssertionError(f"image differs (max {float(np.abs(color - want_color).max())}): " + history)

    try:
        be.initialize_memory(sc)
        for step in range(120):
            op = rs.choice(["render", "render", "render", "snapshot", "burst", "read_slot", "read_slot", "read_image", "statistics",
                            "counters", "clear", "upload", "write", "sync", "kernel_time", "display", "pin", "walk"])
            if op == "walk":
                # the reference's loop for a while: one short call after the other, waiting for each - what the library renders
                # AHEAD of (launches for the next calls, up to four calls each: tests/test_render_ahead_gpu.py) - with the counters or
                # the image checked at a random call; the ops around it leave the pattern wherever they please
                n = int(rs.choice([1, 1, 1, 2, 3]))
                calls = int(rs.randint(2, 14))
                first = int(rs.randint(0, N_IDS - calls * n + 1))
                look = int(rs.randint(0, calls))
                log.append(("walk", first, n, calls))
                for c in range(calls):
                    be.render(first + c * n, n)
                    be.synchronize()
                    for k in range(first + c * n, first + (c + 1) * n):
                        m.add(k)
                    if c == look and rs.rand() < 0.5:
                        assert be.counters() == m.totals, (be.counters(), m.totals, log[-12:])
                    elif c == look:
                        same_image(be.read_image(), m.color, m.count, f"image at call {c} of a walk")
                continue
            if op == "burst" and not bursts:
                with pytest.raises(PtmiError):
                    be.render_snapshots(0, 2, 0)
                continue
            if op == "render":
                first, n = int(rs.randint(0, N_IDS - 6)), int(rs.choice([1, 1, 2, 3, 5]))
                log.append(("render", first, n))
                be.render(first, n)
                for k in range(first, first + n):
                    m.add(k)
            elif op == "snapshot":
                slot = int(rs.randint(0, few))
                log.append(("snapshot", slot))
                be.snapshot(slot)
                m.slots[slot] = (m.color.copy(), m.count.copy())
            elif op == "burst":
                first, n, slot = int(rs.randint(0, N_IDS - 8)), int(rs.randint(1, 8)), int(rs.randint(0, few))
                log.append(("burst", first, n, slot))
                be.render_snapshots(first, n, slot)
                for k in range(n):
                    m.add(first + k)
                    m.slots[(slot + k) % ring] = (m.color.copy(), m.count.copy())
            elif op == "read_slot":
                if m.slots:
                    slot = int(rs.choice(sorted(m.slots)))
                    log.append(("read_slot", slot))
                    same_image(be.read_snapshot(slot, out=pinned if rs.rand() < 0.5 else None), *m.slots[slot], f"slot {slot}")
                else:
                    with pytest.raises(PtmiError):
                        be.read_snapshot(int(rs.randint(0, ring)))
            elif op == "read_image":
                log.append(("read_image",))
                same_image(be.read_image(out=pinned if rs.rand() < 0.5 else None), m.color, m.count, "image")
            elif op == "statistics":
                log.append(("statistics",))
                got = be.read_statistics()
                assert all(np.array_equal(a.astype(np.int64), b) for a, b in zip(got, m.stats)), log[-12:]
            elif op == "counters":
                log.append(("counters",))
                got = be.counters()
                assert got == (m.totals or {k: 0 for k in got}), (got, m.totals, log[-12:])
            elif op == "clear":
                log.append(("clear",))
                be.clear()
                m.reset()
            elif op == "upload":
                name = SCENES[int(rs.randint(0, len(SCENES)))]
                log.append(("upload", name))
                sc, per_it = per_iteration[name, da]
                be.initialize_memory(sc)
                assert (be.literal_kernel_reason() is not None) == (name.startswith("fuzz") and "h" in name.split("_")[0])
                m = Model(per_it)
            elif op == "write":
                log.append(("write",))
                color = rs.uniform(0, 4, (H, W, 4)).astype(np.float32)
                count = np.full((H, W), float(rs.randint(0, 9)), np.float32)
                be.write_image(color, count)
                m.color, m.count = color.copy(), count.copy()
            elif op == "pin":
                log.append(("pin", not is_pinned))
                for a in pinned:
                    (be.unpin_host_buffer if is_pinned else be.pin_host_buffer)(a)
                is_pinned = not is_pinned
            elif op == "sync":
                be.synchronize()
            elif op == "kernel_time":
                ms, launches = be.kernel_time()
                assert ms >= 0 and launches >= 0
            else:
                log.append(("display",))
                got = be.read_display()
                from opencl_pathtracer_amd import output
                want = output.to_display_rgb(m.color, m.count)
                if exact:
                    assert np.array_equal(got[:, :3 * W].reshape(H, W, 3)[..., ::-1], want), log[-12:]
        same_image(be.read_image(), m.color, m.count, "final image")
    finally:
        be.release()


@pytest.mark.parametrize("seed", range(8))
def test_random_call_sequences_super_sampling(seed, per_iteration):
    """The same for an adaptive (SUPER_SAMPLING) context, whose every iteration depends on the accumulators and the variance
    image it finds: renders of arbitrary iteration ids, variance reads, saved states written back (resume), snapshots, clears,
    re-uploads.  The model is the oracle run statefully on the same arrays.  Bit for bit."""
    rs = np.random.RandomState(99 + seed)
    da = seed % 2 == 0
    name = ["cornell", "fuzz3_l1"][seed % 4 == 3]
    sc = per_iteration[name, da][0]
    be = Backend().setup_context(W, H, D, sc.lightsSize, S.JITTERED, super_sampling=True, flags=backend.FLAG_DEFAULT_ARITHMETIC if da else 0)

    def fresh():
        return {"color": np.zeros((H, W, 4), np.float32), "count": np.zeros((H, W), np.float32), "v": np.zeros((H, W, 4), np.float32),
                "stats": (np.zeros(D + 1, np.uint32), np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32), np.zeros(S.MAX_INTERSETCION_NUMBER, np.uint32)),
                "totals": None}

    m, saved, slots, log = fresh(), None, {}, []

    def check(cond, what):
        if not cond:
            raise AssertionError(what + "; calls so far: ... " + " ".join(str(c) for c in log[-40:]))

    try:
        be.initialize_memory(sc)
        for step in range(70):
            op = rs.choice(["render", "render", "render", "read_image", "variance", "statistics", "counters", "clear", "save", "resume",
                            "snapshot", "read_slot", "upload"])
            if op == "render":
                first, n = int(rs.randint(0, 40)), int(rs.choice([1, 2, 4, 9]))
                log.append(("render", first, n))
                be.render(first, n)
                _, _, _, tot = O.oracle_render(sc, W, H, D, n, first_iteration=first, super_sampling=True, default_arithmetic=da,
                                               first_sample_guard=True, into=(m["color"], m["count"], m["stats"], None), image_v=m["v"])
                m["totals"] = dict(tot) if m["totals"] is None else {k: m["totals"][k] + tot[k] for k in tot}
            elif op == "read_image":
                log.append(("read_image",))
                color, count = be.read_image()
                check(np.array_equal(count, m["count"]), "sample counts differ")
                check(np.array_equal(color.view(np.uint32), m["color"].view(np.uint32)), "image bits differ")
            elif op == "variance":
                log.append(("variance",))
                check(np.array_equal(be.read_variance().view(np.uint32), m["v"].view(np.uint32)), "variance image bits differ")
            elif op == "statistics":
                log.append(("statistics",))
                check(all(np.array_equal(a, b) for a, b in zip(be.read_statistics(), m["stats"])), "histograms differ")
            elif op == "counters":
                got = be.counters()
                check(got == (m["totals"] or {k: 0 for k in got}), f"counters {got} instead of {m['totals']}")
            elif op == "clear":
                log.append(("clear",))
                be.clear()
                m = fresh()
            elif op == "save":
                log.append(("save",))
                saved = (m["color"].copy(), m["count"].copy(), m["v"].copy())
            elif op == "resume" and saved is not None:
                log.append(("resume",))
                be.write_image(saved[0], saved[1])
                be.write_variance(saved[2])
                m["color"], m["count"], m["v"] = saved[0].copy(), saved[1].copy(), saved[2].copy()
            elif op == "snapshot":
                slot = int(rs.randint(0, 5))
                log.append(("snapshot", slot))
                be.snapshot(slot)
                slots[slot] = (m["color"].copy(), m["count"].copy())
            elif op == "read_slot" and slots:
                slot = int(rs.choice(sorted(slots)))
                log.append(("read_slot", slot))
                color, count = be.read_snapshot(slot)
                check(np.array_equal(count, slots[slot][1]) and np.array_equal(color.view(np.uint32), slots[slot][0].view(np.uint32)), f"slot {slot} differs")
            elif op == "upload":
                log.append(("upload",))
                be.initialize_memory(sc)
                m, slots = fresh(), {}
    finally:
        be.release()


def test_contexts_driven_from_concurrent_threads(per_iteration):
    """Contexts share nothing a caller can see: four of them - different scenes, arithmetics and kernels - driven at the same time
    from four host threads (ctypes releases the interpreter lock inside the calls) produce what each produces alone."""
    import threading
    jobs = [("cornell", True, 0), ("fuzz3_l1", False, 0), ("fuzz5h_l1", True, 0), ("cornell", False, backend.FLAG_MEGAKERNEL)]
    results, errors = {}, []

    def drive(i, name, da, extra):
        try:
            sc, per_it = per_iteration[name, da]
            be = Backend().setup_context(W, H, D, sc.lightsSize, S.JITTERED, flags=(backend.FLAG_DEFAULT_ARITHMETIC if da else 0) | extra)
            try:
                be.initialize_memory(sc)
                for rep in range(6):
                    be.render(rep * 5, 5)
                    if rep % 2:
                        be.read_image()
                color, count = be.read_image()
                results[i] = (color.copy(), count.copy(), be.counters())
            finally:
                be.release()
        except Exception as e:  # noqa: BLE001 (reported by the main thread)
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=drive, args=(i, *job)) for i, job in enumerate(jobs)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for i, (name, da, extra) in enumerate(jobs):
        sc, per_it = per_iteration[name, da]
        m = Model(per_it)
        for k in range(30):
            m.add(k)
        color, count, counters = results[i]
        assert np.array_equal(count, m.count) and np.array_equal(color.view(np.uint32), m.color.view(np.uint32)), name
        assert counters == m.totals, name
