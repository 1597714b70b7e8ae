#!/usr/bin/env python3
"""Many samples on many fuzzed scenes: the integrator against the reference kernel run beside it, bit for bit.  (GPU box; needs
oracle/_ref.)  What a 24-spp test cannot see - an event once in a million paths, like the refraction that makes a ray NaN - shows
in millions of paths per scene.
usage: tools/fuzz_soak.py [FIRST_SEED [N_SEEDS [SPP]]]  > profiles/r03_fuzz_soak.json   (progress on stderr;
SOAK_PARTIAL=file: the same JSON rewritten after every scene, for runs under a time limit)"""
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create, render_scene  # noqa: E402

from opencl_pathtracer_amd import structs as S  # noqa: E402

# SOAK_MODE=fullsize: 1920 x 1080, depth 10, one light, default build only;  SOAK_MODE=uniform: the UNIFORM sampler's code objects; SOAK_MODE=ss: SUPER_SAMPLING (one light, no hostile records)
MODE = os.environ.get("SOAK_MODE", "")
SPECS = {"": {1: ("feat_64x64_d8", 64, 64, 8), 3: ("matmix_96x96_d8", 96, 96, 8)},
         "uniform": {1: ("cornell_64x48_d4_uni", 64, 48, 4), 3: ("matmix_96x96_d8_uni", 96, 96, 8)},
         "ss": {1: ("cornell_64x48_d4_ss", 64, 48, 4), 3: ("cornell_64x48_d4_ss", 64, 48, 4)},
         "fullsize": {1: ("tris1m_1920x1080_d10", 1920, 1080, 10), 3: ("tris1m_1920x1080_d10", 1920, 1080, 10)}}[MODE]
SAMPLER = S.UNIFORM if MODE == "uniform" else S.JITTERED


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    spp = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    warnings.simplefilter("ignore")
    out = {"spp": spp, "mode": MODE or "jittered", "scenes": [], "paths": 0, "mismatches": []}
    t0 = time.time()
    for seed in range(first, first + n):
        for suffix in ("", "h", "r", "hr", "t"):
            n_lights = 1 if (seed + len(suffix)) % 2 or MODE in ("ss", "fullsize") else 3
            if MODE == "ss" and "h" in suffix:
                continue
            case, w, h, d = SPECS[n_lights]
            tree = suffix == "t"
            name = f"fuzz{seed}{'' if tree else suffix}_l{n_lights}"
            sc = bvh_create(scenes.build(name, w, h))
            if tree:
                scenes.corrupt_tree(sc, seed)
            for strict in ((False,) if MODE == "fullsize" else (False, True)):
                flags = 0 if strict else 16
                ours = render_scene(sc, w, h, d, spp, flags=flags, sampler=SAMPLER, super_sampling=MODE == "ss")
                ref = O.ref_gpu_render(case, sc, w, h, d, spp, strict=strict)
                same = (np.array_equal(ours[0].view(np.uint32), ref[0].view(np.uint32)) and np.array_equal(ours[1], ref[1])
                        and all(np.array_equal(a, b) for a, b in zip(ours[2], ref[2])))
                out["paths"] += w * h * spp
                if not same:
                    bad = int((ours[0].view(np.uint32) != ref[0].view(np.uint32)).any(-1).sum())
                    out["mismatches"].append({"scene": sc.name, "build": "strict" if strict else "default", "pixels": bad,
                                              "depths_ours": ours[2][0].tolist(), "depths_reference": ref[2][0].tolist()})
            out["scenes"].append(sc.name)
            print(f"{sc.name}: {len(out['mismatches'])} mismatches so far, {out['paths'] / 1e6:.0f} M paths, {time.time() - t0:.0f} s", file=sys.stderr, flush=True)
            if os.environ.get("SOAK_PARTIAL"):  # a run under a time limit still leaves its figures: rewritten after every scene
                with open(os.environ["SOAK_PARTIAL"], "w") as f:
                    json.dump({**out, "note": "partial: rewritten after every scene"}, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
