"""Where does a systematic difference against the reference kernel come from?  Renders one parity case at a high sample count
with this integrator and with the reference's default and strict builds (oracle/_ref), and reports the signed mean
difference per class of pixels = material of the camera ray's first hit (from the CPU oracle's path trace), so that a
bias shows up next to the feature that produces it while chaotic flips average out.  Diagnostic; runs on the GPU box.

usage: python tools/diag_bias.py [case] [spp]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, render_scene  # noqa: E402

case = sys.argv[1] if len(sys.argv) > 1 else "matmix_96x96_d8"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
name, sampler, w, h, d = cases.CASES[case]
sc = bvh_create(scenes.build(name, w, h))
first_mat = np.full((h, w), -1, np.int32)
for y in range(h):
    for x in range(w):
        b, _ = O.oracle_trace(sc, w, h, d, x, y, 0, sampler)
        if b:
            first_mat[y, x] = b[0].material_id
r, rn, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp)
s, sn, _, _ = O.ref_gpu_render(case, sc, w, h, d, spp, strict=True)
g, gn, _, _ = render_scene(sc, w, h, d, spp, sampler=sampler)
img = lambda c, n: (c[..., :3] / np.maximum(n, 1)[..., None]).astype(np.float64)
R, S_, G = img(r, rn), img(s, sn), img(g, gn)
print(f"{case} at {spp} spp: rms ours-default {np.sqrt(((G - R) ** 2).mean(axis=(0, 1)))}, strict-default {np.sqrt(((S_ - R) ** 2).mean(axis=(0, 1)))}")
mtype = {0: "STANDART", 1: "WATER", 2: "GLASS", 3: "VARNISHED", 4: "METAL"}
for m in sorted(set(first_mat.ravel().tolist())):
    sel = first_mat == m
    if m >= 0:
        mat = sc.materiaux[m]
        label = f"material {m} ({mtype.get(int(mat['type']), '?')}{'' if mat['isSimpleColor'] else ', textured'})"
    else:
        label = "sky"
    mean = R[sel].mean()
    print(f"  first hit {label:38s} {int(sel.sum()):5d} px  mean {mean:.4f}  ours-default mean {np.mean(G[sel] - R[sel]):+.2e} rms {np.sqrt(np.mean((G[sel] - R[sel]) ** 2)):.2e}"
          f"   strict-default mean {np.mean(S_[sel] - R[sel]):+.2e} rms {np.sqrt(np.mean((S_[sel] - R[sel]) ** 2)):.2e}")
d_img = np.abs(G - R).max(axis=-1)
worst = np.argsort(d_img.ravel())[::-1][:12]
for i in worst:
    y, x = divmod(int(i), w)
    print(f"  pixel ({x:3d},{y:3d}) first-hit material {first_mat[y, x]:2d}: ours {G[y, x].round(5)} default {R[y, x].round(5)} strict {S_[y, x].round(5)}")
