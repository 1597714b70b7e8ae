"""Where does the default-arithmetic mode differ from the reference's default build?  Per case / feature scene: pixels whose
1-spp radiance differs (HIP default-arithmetic vs the reference kernel's default build, both on this GPU), how the
oracle's default build sees the same paths, and the first few differing pixels.  Diagnostic for tests/test_reference_default_gpu.py."""
import sys
import numpy as np
sys.path[:0] = [".", "tests"]
import cases
import oracle_ffi as O
from opencl_pathtracer_amd import render_scene, scenes, bvh_create, backend

DA = backend.FLAG_DEFAULT_ARITHMETIC
names = sys.argv[1:] or ["cornell_64x48_d4"] + ["feat:" + f for f in scenes.FEATURES]
for nm in names:
    if nm.startswith("feat:"):
        case, w, h, d = "feat_64x64_d8", 64, 64, 8
        sc = bvh_create(scenes.build("feat_" + nm[5:], w, h))
        sampler = 0
    else:
        case = nm
        if nm in cases.CASES:
            name, sampler, w, h, d = cases.CASES[case]
        else:  # <scene>_<W>x<H>_d<depth>: a diagnostic specialisation of oracle/ref_configs.txt
            import re
            m = re.match(r"(\w+?)_(\d+)x(\d+)_d(\d+)$", nm)
            name, w, h, d, sampler = m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), 0
        sc = bvh_create(scenes.build(name, w, h))
    tot = 0
    first = []
    for it in range(4):
        r, _, (rd, rb, rt), _ = O.ref_gpu_render(case, sc, w, h, d, 1, first_iteration=it)
        g, _, (gd, gb, gt), _ = render_scene(sc, w, h, d, 1, first_iteration=it, sampler=sampler, flags=DA)
        o, _, _, _ = O.oracle_render(sc, w, h, d, 1, first_iteration=it, sampler=sampler, default_arithmetic=True)
        bad = np.argwhere((g.view(np.uint32) != r.view(np.uint32)).any(-1))
        bad_o = np.argwhere((o.view(np.uint32) != g.view(np.uint32)).any(-1))
        tot += len(bad)
        for y, x in bad[:3]:
            first.append((it, int(x), int(y), r[y, x, :3].tolist(), g[y, x, :3].tolist()))
        print(f"{nm} it {it}: {len(bad)} of {w*h} pixels differ from the reference default build; oracle-DA vs HIP-DA: {len(bad_o)}; "
              f"hist diffs depth {int((rd != gd).sum())} bbx {int((rb != gb).sum())} tri {int((rt != gt).sum())}")
    for f in first[:6]:
        print("   ", f)
