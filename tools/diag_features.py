"""One kernel feature at a time (scenes.feature_scene) through the reference kernel (default and strict builds of the
unmodified .cl, oracle/_ref) and through this integrator: which branch of the integrator does a difference against the
reference come from?  Test/diagnostic infrastructure; runs on the GPU box.

usage: python tools/diag_features.py [spp] [out.json]
Per feature: 1-spp sample agreement (flip fraction, bias) ours-vs-default and strict-vs-default, per-channel RMS at `spp`
ours-vs-default and strict-vs-default (the reference's distance to itself), and the ratio of the two.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, render_scene  # noqa: E402

CASE, W, H, D = "feat_64x64_d8", 64, 64, 8
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
out = {}
for f in scenes.FEATURES:
    sc = bvh_create(scenes.build("feat_" + f, W, H))
    r1, _, _, _ = O.ref_gpu_render(CASE, sc, W, H, D, 1)
    s1, _, _, _ = O.ref_gpu_render(CASE, sc, W, H, D, 1, strict=True)
    g1, _, _, _ = render_scene(sc, W, H, D, 1)
    ours1, self1 = cases.sample_agreement(g1, r1), cases.sample_agreement(s1, r1)
    r, rn, (rdep, _, _), _ = O.ref_gpu_render(CASE, sc, W, H, D, spp)
    s, sn, (sdep, _, _), _ = O.ref_gpu_render(CASE, sc, W, H, D, spp, strict=True)
    g, gn, (gdep, _, _), _ = render_scene(sc, W, H, D, spp)
    rms_ours = float(cases.rms_per_channel(g, gn, r, rn).max())
    rms_self = float(cases.rms_per_channel(s, sn, r, rn).max())
    rms_strict = float(cases.rms_per_channel(g, gn, s, sn).max())
    mean = float(r[..., :3].mean() / spp)
    out[f] = {"flip_ours": ours1["flip_fraction"], "flip_self": self1["flip_fraction"], "bias_ours": ours1["bias"], "bias_self": self1["bias"],
              "rms_ours_vs_default": rms_ours, "rms_strict_vs_default": rms_self, "rms_ours_vs_strict": rms_strict,
              "ratio": rms_ours / max(rms_self, 1e-12), "mean_radiance": mean,
              "mean_rel_ours": float((g[..., :3].mean() - r[..., :3].mean()) / r[..., :3].mean()),
              "mean_rel_strict": float((s[..., :3].mean() - r[..., :3].mean()) / r[..., :3].mean()),
              "depth_L1_ours": int(np.abs(gdep.astype(np.int64) - rdep.astype(np.int64)).sum()),
              "depth_L1_self": int(np.abs(sdep.astype(np.int64) - rdep.astype(np.int64)).sum())}
    o = out[f]
    print(f"{f:18s} rms ours/default {rms_ours:.2e}  strict/default {rms_self:.2e}  ours/strict {rms_strict:.2e}  ratio {o['ratio']:.2f}  "
          f"flips {o['flip_ours']:.4f}/{o['flip_self']:.4f}  mean-rel {o['mean_rel_ours']:+.1e}/{o['mean_rel_strict']:+.1e}  depthL1 {o['depth_L1_ours']}/{o['depth_L1_self']}",
          flush=True)
if len(sys.argv) > 2:
    json.dump({"case": CASE, "spp": spp, "features": out}, open(sys.argv[2], "w"), indent=1)
