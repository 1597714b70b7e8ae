#!/usr/bin/env python3
"""Rate of the wavefront kernel's NANSAFE instantiation on scenes whose records yield NaN distances (zero-area triangles as
the importer emits them), next to the same seeds without those records and to the one-path-per-lane kernel that rendered such
scenes as a whole before round 4 (PTMI_LITERAL_KERNEL=1).  Run on the GPU box:
    python tools/nansafe_rate.py > gpurun_out/r04_nansafe_rate.json
VERDICT r03 item 4: done = the hostile scene renders on the wavefront kernel at >= 0.9 x the clean scene's rate."""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import opencl_pathtracer_amd as pt  # noqa: E402
from opencl_pathtracer_amd import backend, scenes  # noqa: E402
from opencl_pathtracer_amd import structs as S  # noqa: E402

W, H, D, SPP, REPS = 1920, 1080, 10, 8, 9
DA = backend.FLAG_DEFAULT_ARITHMETIC


def rate(sc, flags, env=None, spp=None, reps=None):
    spp, reps = spp or SPP, reps or REPS
    print(f"  rate: {sc.name} {len(sc.triangulation)} triangles {env or ''} ...", file=sys.stderr, flush=True)
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        be = backend.Backend().setup_context(W, H, D, sc.lightsSize, S.JITTERED, flags=flags)
        be.initialize_memory(sc)
        be.render(1000, spp)
        be.synchronize()
        c0 = be.counters()
        t0 = time.perf_counter()
        per_launch = []
        for r in range(reps):
            t1 = time.perf_counter()
            be.render(r * spp, spp)
            be.synchronize()
            per_launch.append(time.perf_counter() - t1)
            print(f"    {r + 1}/{reps} after {time.perf_counter() - t0:.1f} s", file=sys.stderr, flush=True)
        dt = time.perf_counter() - t0
        c1 = be.counters()
        st = be.scheduler_stats()
        why = be.literal_kernel_reason()
        be.release()
    finally:
        for k in (env or {}):
            del os.environ[k]
    seg = c1["segments"] - c0["segments"]
    med = sorted(per_launch)[len(per_launch) // 2]
    return {"Msamples/s": seg / dt / 1e6, "Mpaths/s": (c1["paths"] - c0["paths"]) / dt / 1e6, "paths_retraced": st["paths_retraced"],
            # launches differ where a rare path walks the whole tree ten times over (a NaN record as its FINAL hit): seconds, anywhere
            "seconds_per_launch": [round(x, 4) for x in per_launch], "Msamples/s_at_the_median_launch": seg / reps / med / 1e6,
            "paths": c1["paths"], "iterations_per_launch": spp, "launches": reps,
            "literal_kernel_reason": why.decode() if isinstance(why, bytes) else why}


import copy  # noqa: E402

import numpy as np  # noqa: E402


def without_bad_records(sc):
    """The clean twin of a hostile scene: the same scene minus the triangles that make the library choose the NANSAFE
    instantiation (non-finite or astronomically large records, zero-area triangles), with a tree of its own."""
    t = sc.triangulation
    ok = np.ones(len(t), bool)
    for f in ("S1", "S2", "S3"):
        ok &= np.isfinite(t[f]).all(axis=1) & (np.abs(t[f]) <= 2097152.0).all(axis=1)
    for f in ("N", "N1", "N2", "N3"):
        ok &= np.isfinite(t[f]).all(axis=1) & (np.abs(t[f]) <= 16.0).all(axis=1)
    u, v = (t["S2"] - t["S1"]).astype(np.float64), (t["S3"] - t["S1"]).astype(np.float64)
    uv, uu, vv = (u * v).sum(1), (u * u).sum(1), (v * v).sum(1)
    ok &= np.float32(uv * uv - uu * vv) != 0
    twin = copy.copy(sc)
    twin.triangulation = t[ok].copy()
    twin.triangulation["id"] = np.arange(ok.sum(), dtype=np.uint32)
    twin.bvh = None
    return pt.bvh_create(twin), int((~ok).sum())


def with_zero_area_triangles(sc, k, seed=5):
    return pt.bvh_create(scenes.add_zero_area_triangles(sc, k, seed))


def reference_rate(sc):
    """The reference's own kernel (oracle/_ref: unmodified source, its own build options) on the same scene, one iteration after
    a warm-up launch: what the paths that meet a NaN record cost THERE."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    case = f"tris1m_{W}x{H}_d{D}"  # (the code object is specialised on sampler, size, depth and light count only)
    if not O.have_ref_kernel(case) or sc.lightsSize != 1:
        return None
    print(f"  reference kernel: {sc.name} ...", file=sys.stderr, flush=True)
    O.ref_gpu_render(case, sc, W, H, D, 1, first_iteration=7)
    _, _, (dep, _, _), ms = O.ref_gpu_render(case, sc, W, H, D, 1)
    seg = int(sum(min(k + 1, D) * int(n) for k, n in enumerate(dep)))  # (upper bound: a path that ends on a hit at depth k made k queries)
    return {"Mpaths/s": W * H / ms / 1e3, "kernel_ms_per_iteration": ms}


def entry(name, clean, hostile, removed=None):
    e = {"triangles": int(len(hostile.triangulation)),
         "clean_wavefront": rate(clean, DA),
         "hostile_wavefront_nansafe": rate(hostile, DA),
         "hostile_one_path_per_lane": rate(hostile, DA, {"PTMI_LITERAL_KERNEL": "1"}, spp=2, reps=2)}
    if removed is not None:
        e["bad_records"] = removed
    ref = reference_rate(hostile)
    if ref:
        e["reference_kernel_on_the_hostile_scene"] = ref
        e["nansafe_over_reference_kernel"] = e["hostile_wavefront_nansafe"]["Mpaths/s"] / ref["Mpaths/s"]
    e["nansafe_over_clean"] = e["hostile_wavefront_nansafe"]["Msamples/s"] / e["clean_wavefront"]["Msamples/s"]
    e["nansafe_over_clean_at_the_median_launch"] = (e["hostile_wavefront_nansafe"]["Msamples/s_at_the_median_launch"] /
                                                    e["clean_wavefront"]["Msamples/s_at_the_median_launch"])
    e["nansafe_over_one_path_per_lane"] = e["hostile_wavefront_nansafe"]["Msamples/s"] / e["hostile_one_path_per_lane"]["Msamples/s"]
    e["share_of_paths_retraced"] = e["hostile_wavefront_nansafe"]["paths_retraced"] / e["hostile_wavefront_nansafe"]["paths"]
    out["scenes"][name] = e
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_nansafe_rate_partial.json"), "w"), indent=1)  # (a killed run keeps what it has)
    return e


out = {"config": f"{W}x{H}, depth {D}, {SPP} iterations per launch x {REPS}, default arithmetic, JITTERED",
       "note": "clean = the same scene without its bad records (fuzz: scenes.fuzz_scene hostile set removed; tris1m: before the "
               "zero-area triangles were added).  A path that meets a NaN distance is expensive by itself - nothing is 'too far' "
               "for it any more, so it walks most of the tree (the reference's kernel does the same) - so a scene where 7 % of the "
               "paths do (fuzz3h) cannot run at its clean twin's rate; a mesh with a few degenerate triangles does.", "scenes": {}}
for seed in (3, 7):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        hostile = pt.bvh_create(scenes.build(f"fuzz{seed}h_l1", W, H))
        clean, removed = without_bad_records(hostile)
    entry(f"fuzz{seed}h_l1", clean, hostile, removed)
base = pt.bvh_create(scenes.build("tris1m", W, H))
for k in (1, 100):
    entry(f"tris1m + {k} zero-area triangle(s)", base, with_zero_area_triangles(base, k), k)
print(json.dumps(out, indent=1))
