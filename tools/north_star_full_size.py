"""north_star's parity clause at BASELINE's FULL sizes and sample counts, against the kernel the reference's own build line
produces (oracle/_ref/ref_kernel_*.hsaco, OpenCL default arithmetic), run beside the integrator on this GPU:
    configs[2]  ~1M random triangles, 1920x1080, depth 10, 256 spp   (the reference kernel needs ~36 s)
    configs[1]  Cornell box,          1920x1080, depth 8, 1024 spp   (~56 s)
    configs[4]  stand-in (scenes.maya_like), 3840x2160, depth 16, 64 of its 2048 spp   (~17 s)
Per config: per-channel RMS of the default-arithmetic mode (expected 0: the images are equal bit for bit) and of the strict
mode (= the reference's own strict-vs-default distance, tests/test_parity_gpu.py) vs that kernel, equality of counts and
histograms, and both kernels' path rates.  Writes gpurun_out/r04_north_star_full_size.json (copied to profiles/).
usage: python tools/north_star_full_size.py [tris1m|cornell ...] [--spp-scale F]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import Backend, scenes, bvh_create, backend  # noqa: E402

CONFIGS = {"tris1m": ("tris1m_1920x1080_d10", "tris1m", 1920, 1080, 10, 256, "BASELINE configs[2]"),
           "cornell": ("cornell_1920x1080_d8", "cornell", 1920, 1080, 8, 1024, "BASELINE configs[1]"),
           # configs[4]'s stand-in (SURVEY 8d Config 5) at its full size; 64 of its 2048 spp (the reference kernel needs 17 s for them)
           "mayalike": ("mayalike_3840x2160_d16", "mayalike", 3840, 2160, 16, 64, "BASELINE configs[4] stand-in, 64 of 2048 spp")}


def ours(sc, w, h, d, spp, flags):
    be = Backend().setup_context(w, h, d, sc.lightsSize, flags=flags)
    be.initialize_memory(sc)
    t0 = time.perf_counter()
    be.render(0, spp)
    be.synchronize()
    dt = time.perf_counter() - t0
    color, count = be.read_image()
    stats = be.read_statistics()
    be.release()
    return color, count, stats, dt


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(CONFIGS)
    scale = float(sys.argv[sys.argv.index("--spp-scale") + 1]) if "--spp-scale" in sys.argv else 1.0
    out_path = os.path.join(ROOT, "gpurun_out", "r04_north_star_full_size.json")
    data = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for nm in names:
        case, scene, w, h, d, spp, what = CONFIGS[nm]
        spp = max(1, int(spp * scale))
        sc = bvh_create(scenes.build(scene, w, h))
        print(f"{nm}: reference kernel (default build), {spp} spp ...", flush=True)
        r_color, r_count, (r_dep, r_bbx, r_tri), ref_ms = O.ref_gpu_render(case, sc, w, h, d, spp)
        print(f"{nm}: reference kernel {ref_ms / 1e3:.1f} s", flush=True)
        a_color, a_count, (a_dep, a_bbx, a_tri), a_dt = ours(sc, w, h, d, spp, backend.FLAG_DEFAULT_ARITHMETIC)
        s_color, s_count, _, s_dt = ours(sc, w, h, d, spp, 0)
        rms_da = cases.rms_per_channel(a_color, a_count, r_color, r_count)
        rms_strict = cases.rms_per_channel(s_color, s_count, r_color, r_count)
        paths = w * h * spp
        # bounces the reference's SOURCE leaves undefined (ptmi_invariant_checks.refraction_undefined_in_reference: its water
        # material refracting a totally reflected ray along an uninitialised direction): counted by the statistics build over the
        # same iterations - the only paths that may differ from the reference kernel's
        be = Backend().setup_context(w, h, d, sc.lightsSize, flags=backend.FLAG_DEFAULT_ARITHMETIC | backend.FLAG_SCHEDULER_STATS)
        be.initialize_memory(sc)
        be.render(0, spp)
        be.synchronize()
        undefined = be.invariant_checks()["refraction_undefined_in_reference"]
        be.release()
        differing = int((a_color.view(np.uint32) != r_color.view(np.uint32)).any(-1).sum())
        rec = {"config": what, "case": case, "width": w, "height": h, "ray_max_depth": d, "spp": spp,
               "pixels_that_differ": differing, "bounces_undefined_in_the_reference_source": int(undefined),
               "every_difference_accounted_for": bool(differing <= undefined),
               "rms_default_arithmetic_mode_vs_reference_default_build": [float(x) for x in rms_da],
               "image_bits_equal": bool(np.array_equal(a_color.view(np.uint32), r_color.view(np.uint32))),
               "counts_equal": bool(np.array_equal(a_count, r_count)),
               "histograms_equal": bool(np.array_equal(a_dep, r_dep) and np.array_equal(a_bbx, r_bbx) and np.array_equal(a_tri, r_tri)),
               "rms_strict_mode_vs_reference_default_build": [float(x) for x in rms_strict],
               "reference_kernel_seconds": ref_ms / 1e3, "reference_kernel_mpaths_per_s": paths / ref_ms / 1e3,
               "integrator_default_arithmetic_seconds": a_dt, "integrator_default_arithmetic_mpaths_per_s": paths / a_dt / 1e6,
               "integrator_strict_seconds": s_dt, "integrator_strict_mpaths_per_s": paths / s_dt / 1e6}
        print(json.dumps(rec), flush=True)
        data[nm] = rec
        json.dump(data, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
