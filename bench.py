#!/usr/bin/env python3
"""bench.py - the reference's headline metric on MI355X.

Metric (BASELINE.json): Msamples/s, sample = one path segment = one closest-hit query
(BVH_IntersectRay call; "paths x bounces"), at 1920x1080.  Default workload at every N: BASELINE.json
configs[2] -- the synthetic 1M-random-triangle scene, 1920x1080, depth 10, JITTERED, one point light
(generator pinned in opencl_pathtracer_amd/scenes.py, BVH from the bit-compatible builder).
A "step" = one pass of the integrator over one batch = --spp-per-step iterations (samples per pixel)
of the full image.  Inputs (scene + accumulators) are resident in HBM before the timed region; the timed
region ends with the framebuffer on the host of rank 0 (one readback).

Multi-GPU (torchrun, one rank per GPU): iteration ids are partitioned over ranks, each rank renders its ids
on a full scene replica with no data-path exchange, and ONE RCCL reduce of the fused float[5*W*H] accumulators
onto rank 0 closes the timed region.  Weak scaling by default (every rank renders --spp-per-step ids per step);
--strong / --total-spp fix the job instead (BASELINE configs[3]: 4096 spp split over the ranks).

Arithmetic: --arithmetic default (the default) renders in the arithmetic of the kernel the reference's own build line
produces (PTMI_FLAG_DEFAULT_ARITHMETIC: images equal that kernel's bit for bit, tests/test_reference_default_gpu.py);
--arithmetic strict is the other bit-exact mode (the reference's strict build).

Prints ONE JSON line on rank 0.  Extra objects:
  "roofline"          the ceiling that BINDS this kernel (bound / frac <= 1), from the PMC passes of this command committed
                      under profiles/ (used only if they were taken on this very kernel source) and this run's launch time
                      (HIP events on the kernel's stream); beside it "hbm_measured" (memory-side traffic / time) and
                      "hbm_algorithmic_model" (SURVEY 8d's algorithmic bytes / time: what the traversal must READ, most of it
                      from the caches - a model, not a ceiling, and may exceed the HBM peak);
  at N=1 "reference_kernel"  the reference's own Kernel_Main (unmodified source, its own build options, oracle/_ref/*.hsaco)
                      timed on this GPU in this run, and the ratios to it (-> vs_baseline);
         "cpu_baseline"      the CPU oracle = scalar port of the reference kernel on the host cores, bounded sample;
         "boundary"          the integrator driven the way the reference drives its backend: one image per launch, a
                             readback and a callback after every image (OpenCL.cpp:76-107).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth, same guide
N_SIMD = 1024           # 256 CUs x 4 SIMDs
# What the 256 L1s (TCP) of the chip take for this kernel's kind of load - every lane reads ONE 64-byte record of its own with
# dwordx4 loads - comes from a committed record: profiles/r04_l1_gather_microbench.json (tools/microbench/l1_gather.hip run by
# tools/l1_ceiling.sh on MI355X, with the L1's own counters of the same launches beside every rate).  Its `kernel_mix` regime is
# the kernel's own by its PMC profile: 600 of 1000 records cost a line fill, served by the L2s, 3.14 accesses per record; the
# counters show that regime access-bound (it reaches 94 % of the all-hits rate; every record a NEW line reaches 73 %).
L1_ACCESSES_PER_RECORD = 3.14  # what this kernel needs per 64-byte record (PMC: a rejected triangle reads half of its record)
L1_FILL_RATE_L2 = 258.2e9      # line fills from the L2s / the Infinity Cache (tools/microbench/record_size, profiles/r01_ab_late_round.txt)
L1_FILL_RATE_MALL = 58.1e9


def l1_ceiling():
    """(accesses per second, where the figure comes from)"""
    path = os.path.join(ROOT, "profiles", "r04_l1_gather_microbench.json")
    try:
        rec = json.load(open(path))
        c = rec["ceiling"]
        return c["kernel_mix_G_per_s"] * 1e9, {"source": "profiles/r04_l1_gather_microbench.json (tools/l1_ceiling.sh)",
                                               "regime": "kernel_mix: 600 of 1000 records cost a line fill from the L2s, 3.14 accesses per record",
                                               "pure_access_rate_G_per_s": c["pure_access_rate_G_per_s"],
                                               "every_record_a_new_line_from_l2_G_per_s": c.get("every_record_a_new_line_from_l2_G_per_s")}
    except (OSError, KeyError, ValueError):
        # (the record is part of the repository; without it the round-1 figure: 214.5 G records/s x 4 on a 2 MB set)
        return 4 * 214.5e9, {"source": "profiles/r01_ab_late_round.txt (fallback: profiles/r04_l1_gather_microbench.json not found)"}


L1_ACCESS_RATE, L1_CEILING_SOURCE = l1_ceiling()


def algorithmic_bytes(c, n_pixels, n_flush, textured_hits=0):
    """32 B per box test, 48 B per triangle test, 96 B per surface hit (+ 24 B of uv and 4 B of texel per hit on
    a textured material), 40 B per pixel per accumulator flush."""
    return 32 * c["box_tests"] + 48 * c["triangle_tests"] + 96 * c["surface_hits"] + 28 * textured_hits + 40 * n_pixels * n_flush


def textured_hit_fraction(pt, scene, W, H, D, device, flags):
    """Surface hits on a material with a file texture / all surface hits, from ONE iteration of the statistics build of the
    kernel (ptmi_scheduler_stats.textured_hits: the production instantiation does not count them)."""
    if len(scene.textures) == 0 or not (scene.materiaux["isSimpleColor"] == 0).any():
        return 0.0
    from opencl_pathtracer_amd.backend import FLAG_SCHEDULER_STATS
    be = pt.Backend().setup_context(W, H, D, scene.lightsSize, pt.structs.JITTERED, device=device, flags=flags | FLAG_SCHEDULER_STATS)
    be.initialize_memory(scene)
    be.render(0, 1)
    be.synchronize()
    c, s = be.counters(), be.scheduler_stats()
    be.release()
    return s["textured_hits"] / max(c["surface_hits"], 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=32,
                    help="iterations per step: per GPU (weak scaling, default) or for the whole job (--strong)")
    ap.add_argument("--strong", action="store_true", help="strong scaling: --spp-per-step ids per step are split over the ranks")
    ap.add_argument("--total-spp", type=int, default=0,
                    help="strong scaling of a fixed job (BASELINE configs[3]: 4096): steps = total / spp-per-step, no warm-up")
    ap.add_argument("--scene", default="tris1m")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--kernel", choices=["wavefront", "megakernel"], default="wavefront")
    ap.add_argument("--arithmetic", choices=["default", "strict"], default="default",
                    help="default = the arithmetic of the reference's own build (OpenCL default: what its build line produces); "
                         "strict = its -ffp-contract=off -cl-fp32-correctly-rounded-divide-sqrt build.  Both bit-exact modes")
    ap.add_argument("--no-reference-kernel", action="store_true", help="skip timing the reference's own kernel beside (N=1 only)")
    ap.add_argument("--scheduler-stats", action="store_true", help="also report wave-scheduler statistics (costs ~1 %)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-boundary", action="store_true", help="skip the reference-style per-image loop (N=1 only)")
    ap.add_argument("--no-histograms", action="store_true",
                    help="skip the reference's three per-path statistics atomics (FullKernel.cl:1319-1331); default: keep them")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo is a rehearsal of the N>1 control flow on a one-GPU box "
                         "(all ranks share GPU 0, accumulators are reduced through host memory)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="image rows the CPU baseline renders (0 = auto)")
    ap.add_argument("--cpu-spp", type=int, default=0, help="iterations the CPU baseline renders (0 = auto: 10-30 s of CPU work)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import opencl_pathtracer_amd as pt
    from opencl_pathtracer_amd.backend import FLAG_NO_HISTOGRAMS, FLAG_MEGAKERNEL, FLAG_SCHEDULER_STATS, FLAG_DEFAULT_ARITHMETIC
    from opencl_pathtracer_amd.distributed import FusedAccumulators, shard_iterations

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
    strong = args.strong or args.total_spp > 0
    if args.total_spp > 0:
        args.steps, args.warmup = max(1, args.total_spp // B), 0
    t0 = time.time()
    scene = pt.bvh_create(pt.scenes.build(args.scene, W, H))
    t_scene = time.time() - t0

    flags = ((FLAG_NO_HISTOGRAMS if args.no_histograms else 0) | (FLAG_MEGAKERNEL if args.kernel == "megakernel" else 0)
             | (FLAG_SCHEDULER_STATS if args.scheduler_stats else 0) | (FLAG_DEFAULT_ARITHMETIC if args.arithmetic == "default" else 0))
    be = pt.Backend().setup_context(W, H, D, scene.lightsSize, pt.structs.JITTERED, device=local_rank, flags=flags)
    be.initialize_memory(scene)
    fb = FusedAccumulators(W, H, device)
    fb.bind(be)
    # The integrator keeps its own streams (consecutive launches alternate between two of them, so that the ramp-up of one
    # fills the CUs the ragged end of the other leaves idle); the collective and the readback below run on a torch stream
    # and are ordered behind the last launch by FusedAccumulators.reduce_to, which waits for the integrator first.
    stream = torch.cuda.Stream(device)
    host_image = torch.empty(5 * W * H, dtype=torch.float32, pin_memory=True) if rank == 0 else None

    def step(s):
        if strong:  # global step s covers ids [s*B, (s+1)*B): this rank's contiguous share of them
            first, n = shard_iterations(s * B, B, rank, world)
        else:       # weak: ids [s*B*world, (s+1)*B*world), a block of B per rank
            first, n = (s * world + rank) * B, B
        if n:
            be.render(first, n)

    torch.cuda.synchronize(device)
    torch.cuda.set_stream(stream)
    for s in range(args.warmup):
        step(s)
    if world > 1 and args.backend == "nccl":
        # warm the collective too (RCCL sets up its channels for a message size on first use): same size, same
        # stream, scratch data - the accumulators are only reduced once, inside the timed region
        scratch = torch.zeros_like(fb.buffer)
        dist.reduce(scratch, dst=0, op=dist.ReduceOp.SUM)
        del scratch
    if rank == 0 and args.warmup > 0:
        # ... and the readback path (torch sets up its device-to-host copy on first use: ~7 ms, half of a two-launch Cornell region)
        host_image.copy_(fb.buffer, non_blocking=True)
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
        be.synchronize()  # the launches run on the integrator's own streams
        host = fb.buffer.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        fb.buffer.copy_(host)
    else:
        fb.reduce_to(0)  # waits for this rank's launches, then the one collective of a sharded render: RCCL reduce over xGMI (no-op at N=1)
    if rank == 0:
        host_image.copy_(fb.buffer, non_blocking=True)  # t_render ends with the framebuffer on the host
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
    be.release()

    if rank == 0:
        n_pix = W * H
        color = host_image[: 4 * n_pix].view(H, W, 4).numpy()
        count = host_image[4 * n_pix:].view(H, W).numpy()
        spp_done = (args.warmup + args.steps) * B * (1 if strong else world)
        assert np.isfinite(color).all() and float(count.min()) == float(spp_done) == float(count.max()), \
            f"sample count {count.min()}..{count.max()} != {spp_done}"
        tex_frac = textured_hit_fraction(pt, scene, W, H, D, local_rank, flags) if args.kernel == "wavefront" else 0.0
        n_texel = tex_frac * delta["surface_hits"]
        b_alg = algorithmic_bytes(delta, n_pix, launches, n_texel)
        avg_launch_s = kernel_ms / 1e3 / max(launches, 1)
        achieved = b_alg / max(launches, 1) / avg_launch_s / 1e9
        records_per_launch = (delta["box_tests"] / 2 + delta["triangle_tests"]) / max(launches, 1)
        model = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_hbm_peak": achieved / HBM_PEAK_GBS,
                 "algorithmic_bytes_per_launch": b_alg / max(launches, 1),
                 "textured_hits_per_surface_hit": tex_frac, "texel_lookups_per_launch": n_texel / max(launches, 1),
                 "note": "SURVEY 8d ALGORITHMIC bytes (32 B per box test, 48 B per triangle test, 96 B per surface hit, 24 B of uv + "
                         "4 B of texel per hit on a textured material, 40 B per pixel flush) / launch time: what the traversal must read, nearly all of it served by the L1s / L2s / "
                         "Infinity Cache (hbm_measured is what reaches memory) - a model of the work, NOT a ceiling of this kernel"}
        roof = {"bound": None, "achieved": None, "peak": None, "unit": None, "frac": None, "traffic": None,
                "kernel": "render_wavefront_kernel" if args.kernel == "wavefront" else "render_kernel",
                "arithmetic": args.arithmetic,
                "launches": launches, "avg_launch_ms": avg_launch_s * 1e3,
                "box_tests_per_path": delta["box_tests"] / max(delta["paths"], 1),
                "triangle_tests_per_path": delta["triangle_tests"] / max(delta["paths"], 1),
                "record_fetches_per_s": records_per_launch / avg_launch_s,
                "hbm_algorithmic_model": model}
        pmc = committed_pmc(args, W, H, D, B)
        if pmc:
            roof.update(binding_ceilings(pmc, avg_launch_s, records_per_launch))
        else:
            # no PMC passes of THIS kernel source on this workload: the one ceiling that can be priced from this run alone -
            # 64-byte record gathers against what the chip's L1s deliver for that access pattern (tools/microbench/record_fetch)
            rate = records_per_launch / avg_launch_s * L1_ACCESSES_PER_RECORD
            roof.update({"bound": "l1_accesses (estimated)", "achieved": rate / 1e9, "peak": L1_ACCESS_RATE / 1e9, "unit": "G accesses/s",
                         "frac": rate / L1_ACCESS_RATE, "ceiling": L1_CEILING_SOURCE,
                         "pmc_fallback": "LOUD: no PMC passes of this kernel source are committed for this workload - the counter-backed "
                                         "table (binding, traffic, hbm_measured) is missing from this line; run tools/profile_round.sh",
                         "note": "no committed PMC passes match this kernel source + workload + arithmetic (profiles/r04_pmc_*.json): the L1 "
                                 f"access rate is ESTIMATED as records fetched x {L1_ACCESSES_PER_RECORD} accesses per record (measured for this "
                                 "kernel on the 1M-triangle workload) against the gather ceiling of tools/microbench/record_fetch; run "
                                 "tools/profile_round.sh for the measured table"})
        out = {
            "metric": "Msamples/s (paths x bounces) at 1920x1080",
            "value": total["segments"] / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
            "arithmetic": args.arithmetic,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload_name(args.scene)}: {len(scene.triangulation)} triangles "
                                   f"({len(scene.bvh)} BVH nodes, depth {scene.bvhMaxDepth}), {W}x{H}, "
                                   f"ray depth {D}, JITTERED, {scene.lightsSize} light(s)",
                       "spp_per_step": B if strong else B * world, "spp_per_step_per_gpu": B / world if strong else B,
                       "spp_total": args.steps * B * (1 if strong else world),
                       "parallelism": f"spp-shard x{world}" if world > 1 else "single GPU",
                       # what the collective of the timed region saw (for the driver's SCALE record): ranks of the process
                       # group the reduce ran in, its backend, and the devices the ranks hold
                       "collective": {"ranks": dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1,
                                      "backend": (dist.get_backend() if (world > 1 and dist.is_initialized()) else None),
                                      "op": "reduce(sum) of float[5*W*H] onto rank 0" if world > 1 else None,
                                      "rank0_device": torch.cuda.get_device_name(device)},
                       "timed_region": "launches + RCCL reduce + one framebuffer readback to rank 0 (pinned host memory)",
                       "scene_build_s": round(t_scene, 2)},
            "Mpaths/s": total["paths"] / elapsed / 1e6,
            "Mshadow_rays/s": total["shadow_rays"] / elapsed / 1e6,
            "segments_per_path": total["segments"] / max(total["paths"], 1),
            # paths whose ray was not a number at some bounce: given up by the launch and traced again by the literal loops (rank 0)
            "paths_retraced": sched["paths_retraced"],
            "roofline": roof,
        }
        if sched["trips_node"]:
            out["wave_scheduler"] = {k: round(sched["lanes_" + k] / (64.0 * sched["trips_" + k]), 3) if sched["trips_" + k] else None
                                     for k in ("node", "triangle", "path")}
            tot = sum(sched["trips_" + k] for k in ("node", "triangle", "path"))
            out["wave_scheduler"]["trip_share"] = {k: round(sched["trips_" + k] / tot, 3) for k in ("node", "triangle", "path")}
            out["wave_scheduler"]["wave_time_in_path_logic"] = round(sched["cycles_path"] / max(sched["cycles_loop"], 1), 3)
            # raw counts per launch (tools/valu_cost_model.py weights the static instruction mix of each trip kind with them)
            out["wave_scheduler"]["trips_per_launch"] = {k: sched["trips_" + k] / max(launches + args.warmup, 1) for k in ("node", "triangle", "path")}
            out["wave_scheduler"]["leaf_item_violations"] = sched["leaf_item_violations"]
        if world == 1 and not args.no_boundary:
            out["boundary"] = boundary_loop(pt, scene, W, H, D, local_rank, flags, out["value"])
        if world == 1 and not args.no_reference_kernel and args.kernel == "wavefront":
            ref = reference_kernel_leg(pt, scene, args, W, H, D, local_rank, flags, out["Mpaths/s"])
            if ref:
                out["reference_kernel"] = ref
            if ref and "ratio_at_the_integrators_launch_size" in ref:
                out["vs_baseline"] = ref["ratio_at_the_integrators_launch_size"]
                out["vs_baseline_note"] = ("value / the reference's own OpenCL kernel (unmodified source, its own build options, 8x8 "
                                           "work-groups) timed on this GPU in this run on the same workload; BASELINE.md holds no published number")
        if "boundary" in out and "Mpaths/s" in out.get("reference_kernel", {}):
            # the per-image protocol as the shim serves it (16 images share a launch, EVERY image read back and shown) against
            # the reference kernel's bare launches (no readback): what a caller of the reference API gets
            seg_per_path = out["segments_per_path"]
            out["reference_kernel"]["ratio_per_image_protocol_shim_default"] = (
                out["boundary"]["per_image_burst16_Msamples/s"] / seg_per_path / out["reference_kernel"]["Mpaths/s"])
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, W, H, D, args.cpu_rows, args.cpu_spp, args.arithmetic == "default")
        print(json.dumps(out), flush=True)

    if world > 1:
        dist.destroy_process_group()


def workload_name(scene):
    return {"tris1m": "BASELINE configs[2]: synthetic random-triangle scene (numpy MT19937 seed 12345)",
            "tris4m": "larger variant of configs[2]: 4M random triangles, 427 MB of records (> the 256 MiB Infinity Cache; still not "
                      "HBM-bound: measured memory-side traffic is ~0.2 % of the HBM peak)",
            "cornell": "BASELINE configs[0]/[1]: Cornell box (point light under a lamp quad)",
            "mayalike": "BASELINE configs[4] stand-in (SURVEY 8d Config 5; no Maya SDK or asset exists): a modelled outdoor set as "
                        "PathTracerMayaImporter would hand it over - tessellated terrain, spheres, torus, pond, walls; Lambert / Phong "
                        "materials + glass, water, metal; four 1024x1024 RGBA file textures, 6 x 512x512 cube-map sky",
            "matmix": "toy material mix (every kernel branch in 1,174 triangles; NOT a BASELINE workload)"}.get(scene, scene)


def kernel_source_digest():
    """SHA-256 over the integrator's device sources: PMC passes are only quoted for the kernel they were taken on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "opencl_pathtracer_amd", "csrc")
    for f in ("kernel_wavefront.hip", "ptmi_device.hpp", "ptmi_shading.hpp", "ptmi_internal.h"):
        h.update(open(os.path.join(d, f), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "ptmi_detmath.h"), "rb").read())
    return h.hexdigest()[:16]


def committed_pmc(args, W, H, D, B):
    """Per-launch means of the rocprofv3 --pmc passes of THIS command line (tools/profile_round.sh: one counter group per
    run, summarised by tools/summarize_pmc.py), committed as profiles/r04_pmc_<scene>_<arithmetic>.json.  bench.py itself
    cannot read PMC counters; the file is used only for the configuration AND the kernel source it was measured on."""
    # (the newest round's passes first; older ones are only quoted if they were taken on this very kernel source, below)
    path = next((p for p in (os.path.join(ROOT, "profiles", f"{r}_pmc_{args.scene}_{args.arithmetic}.json") for r in ("r04", "r03"))
                 if os.path.exists(p)), None)
    if path is None or args.kernel != "wavefront":
        return None
    pmc = json.load(open(path))
    if pmc.get("_config") != {"scene": args.scene, "width": W, "height": H, "depth": D, "arithmetic": args.arithmetic}:
        return None
    # counters are per kernel launch: only comparable when every step of this run is ONE launch of the same size (the library
    # cuts a ptmi_render call into launches of at most 32 iterations, fewer for very large images: ptmi_setup_context)
    tiles = ((W + 7) // 8) * ((H + 7) // 8)
    cap = max(1, min(32, (4 << 30) // (W * H * 20), 0xFFFFFFF0 // (tiles * 64)))
    if B > cap or pmc.get("_spp_per_launch") != B:
        return None
    if pmc.get("_kernel_source_digest") != kernel_source_digest():
        return None
    pmc["_path"] = os.path.relpath(path, ROOT)
    mpath = os.path.join(ROOT, "profiles", os.path.basename(path).replace("_pmc_", "_valu_cost_model_"))
    if os.path.exists(mpath):
        model = json.load(open(mpath))
        if model.get("sources", {}).get("kernel_source_digest") == pmc["_kernel_source_digest"]:
            model["_path"] = os.path.relpath(mpath, ROOT)
            pmc["_valu_cost_model"] = model
    return pmc


def binding_ceilings(pmc, launch_s, records_per_launch):
    """The units this kernel loads, each as achieved / peak <= 1, from the committed PMC passes (per-launch means) and THIS
    run's launch time; the largest is the roofline's `bound`.  traffic = memory-side bytes per launch (FETCH_SIZE doubled as
    the gfx950 note of MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads, plus WRITE_SIZE)."""
    v = lambda k: pmc[k]["per_launch_mean"] if k in pmc else None
    out = {"traffic_source": pmc["_path"] + " (PMC passes of this command on this kernel source, committed; not measured in this run)"}
    if v("FETCH_SIZE") is not None and v("WRITE_SIZE") is not None:
        traffic = (2.0 * v("FETCH_SIZE") + v("WRITE_SIZE")) * 1024.0
        out["traffic"] = traffic
        out["hbm_measured"] = {"achieved": traffic / launch_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": traffic / launch_s / 1e9 / HBM_PEAK_GBS}
    # cycles of the launch: GRBM_GUI_ACTIVE is summed over the 8 XCDs; else the 2.4 GHz maximum clock
    cycles = v("GRBM_GUI_ACTIVE") / 8.0 if v("GRBM_GUI_ACTIVE") else launch_s * 2.4e9
    binding = {}
    if v("SQ_INSTS_VALU") is not None:
        # How busy the vector ALUs are = instructions x what an instruction costs its SIMD / cycles.  The chip has no counter for
        # the second factor (SQ_ACTIVE_INST_VALU counts quad-cycles, one per instruction), so it comes from the code:
        # tools/valu_cost_model.py classes every VALU instruction of the three kinds of trip by the cycle costs measured in
        # tools/microbench/pk_rate.hip (2 / 4 / 8), weights the kinds with the trip counts of --scheduler-stats and checks the
        # predicted instruction count against SQ_INSTS_VALU.  One number; the all-2 / all-4 range stays beside it.
        n = v("SQ_INSTS_VALU")
        entry = {"frac_if_all_2_cycles": n * 2.0 / N_SIMD / cycles, "frac_if_all_4_cycles": n * 4.0 / N_SIMD / cycles}
        model = pmc.get("_valu_cost_model")
        if model:
            mean = model["mean_cycles_per_valu_instruction"]
            entry.update({"achieved": n * mean / N_SIMD, "peak": cycles, "unit": "VALU issue cycles per SIMD per launch",
                          "frac": n * mean / N_SIMD / cycles, "mean_cycles_per_instruction": mean,
                          "model": model["_path"], "model_predicted_over_measured_instructions": model["predicted_over_measured"]})
        binding["valu_busy"] = entry
    if v("SQ_INSTS_SALU") is not None:
        # one scalar unit per CU (256 of them), one instruction per cycle at best
        binding["scalar_unit"] = {"achieved": v("SQ_INSTS_SALU") / 256.0, "peak": cycles, "unit": "instructions per CU per launch",
                                  "frac": v("SQ_INSTS_SALU") / 256.0 / cycles}
    if v("TCC_HIT_sum") is not None and v("TCC_MISS_sum") is not None:
        req = (v("TCC_HIT_sum") + v("TCC_MISS_sum")) * 64.0
        binding["l2_requests"] = {"achieved": req / launch_s / 1e9, "peak": L2_PEAK_GBS, "unit": "GB/s",
                                  "frac": req / launch_s / 1e9 / L2_PEAK_GBS,
                                  "hit_rate": v("TCC_HIT_sum") / (v("TCC_HIT_sum") + v("TCC_MISS_sum"))}
        hit = v("TCC_HIT_sum") / (v("TCC_HIT_sum") + v("TCC_MISS_sum"))
        if v("TCP_TOTAL_CACHE_ACCESSES_sum") is not None and v("TCP_TCC_READ_REQ_sum") is not None:
            # the two rates at which the L1s work: lane accesses (one per lane per load instruction for this gather) and
            # line fills, the latter for this scene's split between fills from the L2s and from the Infinity Cache / HBM
            acc, fills = v("TCP_TOTAL_CACHE_ACCESSES_sum"), v("TCP_TCC_READ_REQ_sum")
            binding["l1_accesses"] = {"achieved": acc / launch_s / 1e9, "peak": L1_ACCESS_RATE / 1e9, "unit": "G accesses/s",
                                      "frac": acc / launch_s / L1_ACCESS_RATE, "per_record": acc / records_per_launch,
                                      "fills_per_access": fills / acc, "ceiling": L1_CEILING_SOURCE}
            t_min = fills * (hit / L1_FILL_RATE_L2 + (1.0 - hit) / L1_FILL_RATE_MALL)
            binding["l1_line_fills"] = {"achieved": fills / launch_s / 1e9, "peak": fills / t_min / 1e9, "unit": "G lines/s",
                                        "frac": t_min / launch_s, "per_record": fills / records_per_launch}
    if v("SQ_WAVE_CYCLES") is not None and v("SQ_WAIT_ANY") is not None:
        binding["wave_cycles"] = {"waiting": v("SQ_WAIT_ANY") / v("SQ_WAVE_CYCLES"),
                                  "issue_stalled": (v("SQ_WAIT_INST_ANY") or 0.0) / v("SQ_WAVE_CYCLES"),
                                  "issuing": (v("SQ_ACTIVE_INST_ANY") or 0.0) / v("SQ_WAVE_CYCLES")}
    ranked = [k for k in binding if "frac" in binding[k]]
    if ranked:
        top = max(ranked, key=lambda k: binding[k]["frac"])
        binding["binds"] = top
        out.update({"bound": top, "achieved": binding[top]["achieved"], "peak": binding[top]["peak"], "unit": binding[top]["unit"],
                    "frac": binding[top]["frac"]})
        out["binding"] = binding
    return out


def reference_kernel_leg(pt, scene, args, W, H, D, device, flags, batched_mpaths):
    """north_star: ">= 10x the repo's OpenCL kernel Msamples/s on 1 x MI355X" - the denominator, measured here: the
    reference's own Kernel_Main (oracle/_ref/ref_kernel_<scene>_<W>x<H>_d<D>.hsaco: unmodified source, the options of its
    own build line, compiled by the image's clang for gfx950) on the same scene, launched as OpenCL.cpp:76-107 launches it -
    one W x H launch per iteration, a wait after each - but with 8x8 work-groups (its own 1x1, OpenCL.cpp:72, would idle 63 of
    64 lanes), 2 iterations, HIP events around each launch.  Beside it this integrator made to launch the same way (one
    iteration per launch, a wait after each).  After the timed region; a baseline, never part of `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    try:
        import oracle_ffi as O
    except Exception as e:  # noqa: BLE001
        return {"skipped": f"tests/oracle_ffi.py not importable: {e}"}
    case = f"{args.scene}_{W}x{H}_d{D}"
    if not O.have_ref_kernel(case):
        return {"skipped": f"oracle/_ref/ref_kernel_{case}.hsaco not present (built by `make -C oracle ref` where the reference tree exists)"}
    n_it = 2
    O.ref_gpu_render(case, scene, W, H, D, 1, first_iteration=7)  # warm-up: module load, clocks
    _, _, _, ms = O.ref_gpu_render(case, scene, W, H, D, n_it)
    ref_mpaths = W * H * n_it / ms / 1e3
    # the integrator CALLED the same way: one iteration per call and a wait after every call.  Twice: as the library serves such a
    # caller by default - once it has seen the caller come back for the next image it keeps launches for the next calls in flight,
    # each for up to four of them (ptmi_api.cpp: render_on_device, PTMI_RENDER_AHEAD, PTMI_RENDER_AHEAD_CALLS) - and with that
    # switched off (one launch per call, nothing before the call).  The timed region starts and ends with an idle device
    # (hipDeviceSynchronize), so what ran ahead before it and what is left running ahead after it cancel: as many iterations are
    # rendered inside it as are asked for.
    import torch
    def blocking_loop(ahead):
        os.environ["PTMI_RENDER_AHEAD"] = ahead
        try:
            be = pt.Backend().setup_context(W, H, D, scene.lightsSize, pt.structs.JITTERED, device=device, flags=flags)
            be.initialize_memory(scene)
            for k in range(100, 116):  # warm-up: the caller is seen to come back, the launches ahead reach their full size
                be.render(k, 1)
                be.synchronize()
            torch.cuda.synchronize(device)
            n_own = 32
            t0 = time.perf_counter()
            for k in range(116, 116 + n_own):
                be.render(k, 1)
                be.synchronize()
            torch.cuda.synchronize(device)
            dt = time.perf_counter() - t0
            be.release()
        finally:
            del os.environ["PTMI_RENDER_AHEAD"]
        return W * H * n_own / dt / 1e6
    own_mpaths = blocking_loop("2")
    own_mpaths_no_ahead = blocking_loop("0")
    return {"Mpaths/s": ref_mpaths, "iterations": n_it, "work_group": "8x8", "kernel_ms_per_iteration": ms / n_it,
            "code_object": f"oracle/_ref/ref_kernel_{case}.hsaco (default build: the reference's own options)",
            "integrator_one_iteration_per_call_blocking_Mpaths/s": own_mpaths,
            "ratio_one_iteration_per_call_blocking": own_mpaths / ref_mpaths,
            "integrator_one_iteration_per_launch_Mpaths/s": own_mpaths_no_ahead,
            "ratio_at_equal_launch_counts": own_mpaths_no_ahead / ref_mpaths,
            "ratio_at_the_integrators_launch_size": batched_mpaths / ref_mpaths,
            "note": "same scene, same samples (in --arithmetic default the two kernels' images are equal bit for bit), same GPU, same "
                    "run; paths/s ratio = samples/s ratio (same segments per path)"}


def boundary_loop(pt, scene, W, H, D, device, flags, batched_value):
    """The integrator driven the way the reference drives its backend (OpenCL.cpp:76-107): ONE image per launch, the
    framebuffer read back into the viewer's buffers and a callback after EVERY image - through the same C ABI calls
    csrc/PathTracer_HIP.cpp makes.  Three variants: the reference's blocking sequence (launch, wait, read, callback),
    one launch per image with the next two queued while image k crosses the bus (ptmi_snapshot / ptmi_read_snapshot into
    page-locked buffers), the shim's default (16 images share a launch and each still gets its snapshot, readback and
    callback: ptmi_render_snapshots), and 32 images per launch and callback (PTMI_IMAGES_PER_LAUNCH=32)."""
    import numpy as np
    be = pt.Backend().setup_context(W, H, D, scene.lightsSize, pt.structs.JITTERED, device=device, flags=flags)
    be.initialize_memory(scene)
    out = (np.empty((H, W, 4), np.float32), np.empty((H, W), np.float32))
    be.pin_host_buffer(out[0])
    be.pin_host_buffer(out[1])
    calls = [0]

    def callback():
        calls[0] += 1

    def run(n_steps, batch, lookahead, first):
        slots = lookahead + 1
        c0 = be.counters()
        t0 = time.perf_counter()
        queued = 0
        for s in range(n_steps):
            while queued < n_steps and queued <= s + lookahead:
                be.render(first + queued * batch, batch)
                be.snapshot(queued % slots)
                queued += 1
            be.read_snapshot(s % slots, out=out)
            callback()
        dt = time.perf_counter() - t0
        c1 = be.counters()
        return (c1["segments"] - c0["segments"]) / dt / 1e6

    def run_bursts(n_bursts, burst, first):
        # what csrc/PathTracer_HIP.cpp does by default: `burst` images share a launch, every one of them is still read back
        # and shown (ptmi_render_snapshots), one burst queued ahead
        c0 = be.counters()
        t0 = time.perf_counter()
        queued = 0
        for b in range(n_bursts):
            while queued < n_bursts and queued <= b + 1:
                be.render_snapshots(first + queued * burst, burst, (queued * burst) % 64)
                queued += 1
            for k in range(burst):
                be.read_snapshot((b * burst + k) % 64, out=out)
                callback()
        dt = time.perf_counter() - t0
        c1 = be.counters()
        return (c1["segments"] - c0["segments"]) / dt / 1e6

    run(4, 1, 2, 0)  # warm-up: staging, snapshot slots
    n = 24
    res = {"images": n,
           "per_image_blocking_Msamples/s": run(n, 1, 0, 4),
           "per_image_pipelined_Msamples/s": run(n, 1, 2, 4 + n),
           "per_image_burst16_Msamples/s": run_bursts(3, 16, 4 + 2 * n),
           "batch32_pipelined_Msamples/s": run(3, 32, 2, 52 + 2 * n),
           "callbacks": calls[0],
           "readback_bytes_per_image": W * H * 20,
           "note": "every variant includes the 20 B/pixel readback and the callback; per_image_* read back and show EVERY image "
                   "like the reference's loop: blocking = its launch / wait / read / callback sequence, pipelined = one launch per "
                   "image with two launches queued ahead, burst16 = the shim's default (16 images share a launch, "
                   "ptmi_render_snapshots); batch32 = one callback per 32 images.  'value' above is 32 iterations per launch "
                   "with one readback at the end"}
    res["per_image_vs_batched"] = res["per_image_burst16_Msamples/s"] / batched_value
    be.release()
    return res


def cpu_baseline(scene, W, H, D, rows, n_iter, default_arithmetic=False):
    """The CPU oracle (scalar C port of the reference kernel, test infrastructure; its build in the arithmetic the bench
    renders in) timed on the host cores on a bounded sample of the SAME workload: iterations 0..n-1 of the first `rows` image
    rows, all host threads."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_ffi as O
    cores = host_cores()
    if rows <= 0 or n_iter <= 0:
        # a short probe gives this scene's rate; then aim at ~20 s of CPU work (10-30 s), whole images first.  Three bands
        # (top, middle, bottom of the image): the first rows alone may be all sky, whose paths are the cheapest there are
        band = max(cores, min(H // 3, 2 * cores))
        t0 = time.perf_counter()
        probed = 0
        for row0 in (0, (H - band) // 2, H - band):
            _, _, _, tot = oracle_rows(O, scene, W, H, D, band, cores, 1, default_arithmetic, first_row=row0)
            probed += tot["paths"]
        rate = probed / max(time.perf_counter() - t0, 1e-6)
        target_paths = 20.0 * rate
        if n_iter <= 0:
            n_iter = max(1, min(64, int(target_paths / (W * H) + 0.5)))
        if rows <= 0:
            rows = H if target_paths >= W * H else max(cores, int(target_paths // W))
    t0 = time.perf_counter()
    _, _, _, totals = oracle_rows(O, scene, W, H, D, rows, cores, n_iter, default_arithmetic)
    dt = time.perf_counter() - t0
    # the port next to the reference's own CPU path (its kernel compiled for x86-64, measured once by the survey in the build
    # container; tools/cpu_port_vs_reference.py times the port on the same workload shape there): per-thread speed ratio
    ratio = None
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "r03_cpu_reference_vs_port.json")))
        ratio = {"port_over_reference_per_thread": rec["port_over_reference_one_thread"]["default" if default_arithmetic else "strict"],
                 "source": "profiles/r03_cpu_reference_vs_port.json (build container, 1 thread, 1M triangles 256x144 2 spp depth 10)"}
    except (OSError, KeyError, ValueError):
        pass
    return {"value": totals["segments"] / dt / 1e6, "unit": "Msamples/s", "cores": cores, "kind": "port",
            "port_vs_reference_cpu_path": ratio,
            "arithmetic": "default" if default_arithmetic else "strict",
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


def oracle_rows(O, scene, W, H, D, rows, threads, n_iter=1, default_arithmetic=False, first_row=0):
    """Render iterations 0..n_iter-1 of rows first_row..first_row+rows-1 of the full image: seed and jitter depend on
    (x, y, W, H), so the full-size W and H are kept and only the row loop is cut short."""
    import ctypes as C
    import numpy as np
    lib = O.oracle(default_arithmetic)
    osc = O.OracleScene(scene, W, H, D)
    color = np.zeros((H, W, 4), np.float32)
    count = np.zeros((H, W), np.float32)
    dep = np.zeros(D + 1, np.uint32)
    bbx = np.zeros(5000, np.uint32)
    tri = np.zeros(5000, np.uint32)
    buf = O.PtoBuffers(O._vp(color), O._vp(count), None, O._vp(dep), O._vp(bbx), O._vp(tri))
    tot = O.PtoTotals()
    lib.pto_render_rows.argtypes = [C.POINTER(O.PtoScene), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                    C.POINTER(O.PtoBuffers), C.c_int, C.POINTER(O.PtoTotals)]
    lib.pto_render_rows.restype = None
    lib.pto_render_rows(C.byref(osc.c), 0, n_iter, first_row, first_row + rows, C.byref(buf), threads, C.byref(tot))
    return color, count, (dep, bbx, tri), tot.as_dict()


if __name__ == "__main__":
    main()
