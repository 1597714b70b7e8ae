#!/usr/bin/env python3
"""bench.py - the reference's headline metric on MI355X.

Metric (BASELINE.json): Msamples/s, sample = one path segment = one closest-hit query
(BVH_IntersectRay call; "paths x bounces"), at 1920x1080.  Workload at every N: BASELINE.json
configs[2] -- the synthetic 1M-random-triangle scene, 1920x1080, depth 10, JITTERED, one point light
(generator pinned in opencl_pathtracer_amd/scenes.py, BVH from the bit-compatible builder).
A "step" = one pass of the integrator over one batch = --spp-per-step iterations (samples per pixel)
of the full image.  Inputs (scene + accumulators) are resident in HBM before the timed region.

Multi-GPU (torchrun, one rank per GPU): iteration ids are partitioned over ranks (weak scaling: each
rank renders --spp-per-step ids per step on a full scene replica, no data-path exchange) and ONE
RCCL reduce of the fused float[5*W*H] accumulators onto rank 0 closes the timed region.

Prints ONE JSON line on rank 0.  Extra objects: "roofline" (HBM, algorithmic bytes from the kernel's
own exact counters / launch time from HIP events on the kernel's stream) and, at N=1, "cpu_baseline"
(the CPU oracle = scalar port of the reference kernel, timed on the host cores on one iteration).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def algorithmic_bytes(c, n_pixels, n_flush):
    """32 B per box test, 48 B per triangle test, 96 B per surface hit, 40 B per pixel per
    accumulator flush (texel reads: none in this workload)."""
    return 32 * c["box_tests"] + 48 * c["triangle_tests"] + 96 * c["surface_hits"] + 40 * n_pixels * n_flush


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=16)
    ap.add_argument("--scene", default="tris1m")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--kernel", choices=["wavefront", "megakernel"], default="wavefront")
    ap.add_argument("--scheduler-stats", action="store_true", help="also report wave-scheduler statistics (costs ~1 %)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-histograms", action="store_true",
                    help="skip the reference's three per-path statistics atomics (FullKernel.cl:1319-1331); default: keep them")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo is a rehearsal of the N>1 control flow on a one-GPU box "
                         "(all ranks share GPU 0, accumulators are reduced through host memory)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of iteration 0 the CPU baseline renders (0 = auto)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import opencl_pathtracer_amd as pt
    from opencl_pathtracer_amd.backend import FLAG_NO_HISTOGRAMS, FLAG_MEGAKERNEL, FLAG_SCHEDULER_STATS
    from opencl_pathtracer_amd.distributed import FusedAccumulators

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the integrator has no CPU path")
    if args.backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()  # rehearsal: ranks may share a card
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    W, H, D, B = args.width, args.height, args.depth, args.spp_per_step
    t0 = time.time()
    scene = pt.bvh_create(pt.scenes.build(args.scene, W, H))
    t_scene = time.time() - t0

    be = pt.Backend().setup_context(W, H, D, scene.lightsSize, pt.structs.JITTERED, device=local_rank,
                                    flags=(FLAG_NO_HISTOGRAMS if args.no_histograms else 0) | (FLAG_MEGAKERNEL if args.kernel == "megakernel" else 0)
                                    | (FLAG_SCHEDULER_STATS if args.scheduler_stats else 0))
    be.initialize_memory(scene)
    fb = FusedAccumulators(W, H, device)
    fb.bind(be)
    # One explicit stream for the kernels AND the collective: the reduce must be ordered after the last
    # launch.  (torch's default stream has handle 0, which ptmi_set_stream reads as "own stream".)
    stream = torch.cuda.Stream(device)
    assert stream.cuda_stream != 0
    be.set_stream(stream.cuda_stream)

    def step(s):
        # global step s covers iteration ids [s*B*world, (s+1)*B*world); this rank takes its block of B
        be.render((s * world + rank) * B, B)

    torch.cuda.synchronize(device)
    torch.cuda.set_stream(stream)
    for s in range(args.warmup):
        step(s)
    if world > 1 and args.backend == "nccl" and args.warmup > 0:
        # warm the collective too (RCCL sets up its channels for a message size on first use): same size, same
        # stream, scratch data - the accumulators are only reduced once, inside the timed region
        scratch = torch.zeros_like(fb.buffer)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        del scratch
    torch.cuda.synchronize(device)
    be.kernel_time()  # drop warm-up launches
    c0 = be.counters()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t_start = time.perf_counter()
    for s in range(args.warmup, args.warmup + args.steps):
        step(s)
    if args.backend == "gloo" and world > 1:  # rehearsal path: gloo has no device tensors
        host = fb.buffer.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        fb.buffer.copy_(host)
    else:
        fb.reduce_to(0)  # the one collective of a sharded render: RCCL reduce over xGMI (no-op at N=1)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start

    kernel_ms, launches = be.kernel_time()
    c1 = be.counters()
    sched = be.scheduler_stats()
    delta = {k: c1[k] - c0[k] for k in c1}

    # whole-job aggregate: sum the counters, take the max time
    cdev = device if args.backend == "nccl" else torch.device("cpu")
    t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    cnt = torch.tensor([delta[k] for k in sorted(delta)], dtype=torch.int64, device=cdev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    total = dict(zip(sorted(delta), [int(x) for x in cnt.tolist()]))

    if rank == 0:
        color, count = fb.images()
        expected = float((args.warmup + args.steps) * B * world)
        assert np.isfinite(color).all() and float(count.min()) == expected == float(count.max()), \
            f"sample count {count.min()}..{count.max()} != {expected}"
        n_pix = W * H
        b_alg = algorithmic_bytes(delta, n_pix, launches)
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = b_alg / max(launches, 1) / avg_launch_s / 1e9
        traffic = committed_traffic(args, W, H, D, B)
        out = {
            "metric": "Msamples/s (paths x bounces) at 1920x1080",
            "value": total["segments"] / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload_name(args.scene)}: {len(scene.triangulation)} triangles "
                                   f"({len(scene.bvh)} BVH nodes, depth {scene.bvhMaxDepth}), {W}x{H}, "
                                   f"ray depth {D}, JITTERED, {scene.lightsSize} light(s)",
                       "spp_per_step_per_gpu": B, "spp_total": args.steps * B * world,
                       "parallelism": f"spp-shard x{world}" if world > 1 else "single GPU",
                       "scene_build_s": round(t_scene, 2)},
            "Mpaths/s": total["paths"] / elapsed / 1e6,
            "Mshadow_rays/s": total["shadow_rays"] / elapsed / 1e6,
            "segments_per_path": total["segments"] / max(total["paths"], 1),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "render_wavefront_kernel" if args.kernel == "wavefront" else "render_kernel",
                         "launches": launches,
                         "avg_launch_ms": avg_launch_s * 1e3, "algorithmic_bytes_per_launch": b_alg / max(launches, 1),
                         "box_tests_per_path": delta["box_tests"] / max(delta["paths"], 1),
                         "triangle_tests_per_path": delta["triangle_tests"] / max(delta["paths"], 1),
                         "note": "achieved = SURVEY 8d algorithmic bytes / kernel time; a fraction above 1 means the record "
                                 "stream is served by L2 + Infinity Cache (see traffic = measured HBM bytes per launch), "
                                 "DESIGN.md 5 'What binds'"},
        }
        if sched["trips_node"]:
            out["wave_scheduler"] = {k: round(sched["lanes_" + k] / (64.0 * sched["trips_" + k]), 3) if sched["trips_" + k] else None
                                     for k in ("node", "triangle", "path")}
            tot = sum(sched["trips_" + k] for k in ("node", "triangle", "path"))
            out["wave_scheduler"]["trip_share"] = {k: round(sched["trips_" + k] / tot, 3) for k in ("node", "triangle", "path")}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, D, args.cpu_rows)
        print(json.dumps(out), flush=True)

    be.release()
    if world > 1:
        dist.destroy_process_group()


def workload_name(scene):
    return {"tris1m": "BASELINE configs[2]: synthetic random-triangle scene (numpy MT19937 seed 12345)",
            "cornell": "BASELINE configs[1]: Cornell box (point light under a lamp quad)",
            "matmix": "BASELINE configs[4] stand-in: textured multi-material scene (no Maya assets exist)"}.get(scene, scene)


def committed_traffic(args, W, H, D, B):
    """roofline.traffic: memory-side bytes per launch from the PMC passes committed under profiles/ (rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same command; FETCH_SIZE doubled as the gfx950 note of
    MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads).  Only quoted for the configuration it was
    measured on; None otherwise (bench.py itself cannot read PMC counters)."""
    path = os.path.join(ROOT, "profiles", "r01_wavefront_pmc.json")
    if not os.path.exists(path) or (args.scene, W, H, D, B, args.kernel) != ("tris1m", 1920, 1080, 10, 16, "wavefront"):
        return None
    pmc = json.load(open(path))
    return (2.0 * pmc["FETCH_SIZE"]["per_launch_mean"] + pmc["WRITE_SIZE"]["per_launch_mean"]) * 1024.0


def cpu_baseline(scene, W, H, D, rows):
    """The CPU oracle (scalar C port of the reference kernel, test infrastructure) timed on the host cores
    on a bounded sample of the SAME workload: iteration 0 of the first `rows` image rows, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    cores = host_cores()
    n_iter = 1
    if rows <= 0:
        # ~0.04 Mpaths/s/thread on this scene: aim at 10-30 s of CPU work, whole images first
        target_paths = 15 * 0.04e6 * cores
        n_iter = max(1, int(target_paths // (W * H)))
        rows = H if n_iter >= 1 and target_paths >= W * H else max(cores, int(target_paths // W))
    t0 = time.perf_counter()
    _, _, _, totals = oracle_rows(O, scene, W, H, D, rows, cores, n_iter)
    dt = time.perf_counter() - t0
    return {"value": totals["segments"] / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "sample": f"iterations 0..{n_iter - 1} of rows 0..{rows - 1} of the {W}x{H} image ({n_iter * rows * W} paths, "
                      f"{totals['segments']} segments) in {dt:.1f} s on {cores} threads",
            "Mpaths/s": totals["paths"] / dt / 1e6}


def host_cores():
    """CPU cores this process may really use: the affinity mask, cut down to the cgroup CPU quota if one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def oracle_rows(O, scene, W, H, D, rows, threads, n_iter=1):
    """Render iteration 0 for the first `rows` rows by running the oracle on a W x rows 'image' whose
    camera rays equal those of rows 0..rows-1 of the full image: seed and jitter depend on (x, y, W, H),
    so the full-size H is kept and only the row loop is cut short."""
    import ctypes as C
    import numpy as np
    lib = O.oracle()
    osc = O.OracleScene(scene, W, H, D)
    color = np.zeros((H, W, 4), np.float32)
    count = np.zeros((H, W), np.float32)
    imgv = np.zeros((1,), np.float32)
    dep = np.zeros(D + 1, np.uint32)
    bbx = np.zeros(5000, np.uint32)
    tri = np.zeros(5000, np.uint32)
    buf = O.PtoBuffers(O._vp(color), O._vp(count), None, O._vp(dep), O._vp(bbx), O._vp(tri))
    tot = O.PtoTotals()
    lib.pto_render_rows.argtypes = [C.POINTER(O.PtoScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(O.PtoBuffers), C.c_int, C.POINTER(O.PtoTotals)]
    lib.pto_render_rows.restype = None
    lib.pto_render_rows(C.byref(osc.c), 0, n_iter, 0, rows, C.byref(buf), threads, C.byref(tot))
    return color, count, (dep, bbx, tri), tot.as_dict()


if __name__ == "__main__":
    main()
