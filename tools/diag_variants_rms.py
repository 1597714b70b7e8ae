"""RMS of library variants (opencl_pathtracer_amd/lib/variants/libptmi_<name>.so) against the reference's default and strict
builds on one parity case at one sample count.  Each variant runs in a child process (the library is chosen at import).
usage: python tools/diag_variants_rms.py case spp name [name ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import numpy as np
    import cases
    import oracle_ffi as O
    from opencl_pathtracer_amd import scenes, bvh_create, render_scene
    case, spp = sys.argv[2], int(sys.argv[3])
    name, sampler, w, h, d = cases.CASES[case]
    sc = bvh_create(scenes.build(name, w, h))
    g, gn, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler)
    r, rn, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp)
    s, sn, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp, strict=True)
    idx = np.arange(w)[None, :] + np.arange(h)[:, None] * w
    deg = (idx % 32) == 0  # pixels whose LCG seed has >= 10 trailing zero bits for every iteration when 32 | W*H
    ia = lambda c, n: (c[..., :3] / np.maximum(n, 1)[..., None]).astype(np.float64)
    G, R, S_ = ia(g, gn), ia(r, rn), ia(s, sn)
    rms = lambda a, b, m: float(np.sqrt(((a - b)[m] ** 2).mean()))
    allpx = np.ones_like(deg)
    print(json.dumps({"ours_vs_default": rms(G, R, allpx), "ours_vs_strict": rms(G, S_, allpx), "strict_vs_default": rms(S_, R, allpx),
                      "ours_vs_default_without_degenerate_seed_pixels": rms(G, R, ~deg), "strict_vs_default_without": rms(S_, R, ~deg),
                      "ours_vs_default_degenerate_only": rms(G, R, deg), "strict_vs_default_degenerate_only": rms(S_, R, deg)}))
    sys.exit(0)
case, spp = sys.argv[1], sys.argv[2]
for name in sys.argv[3:]:
    env = dict(os.environ, PTMI_LIBRARY=os.path.join(ROOT, "opencl_pathtracer_amd", "lib", "variants", f"libptmi_{name}.so"))
    out = subprocess.run([sys.executable, __file__, "--child", case, spp], env=env, capture_output=True, text=True)
    print(name, case, spp, out.stdout.strip() or out.stderr[-400:], flush=True)
