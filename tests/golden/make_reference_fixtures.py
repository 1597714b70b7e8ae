"""Generates tests/golden/ref_<case>.npz by running the REFERENCE kernel on a GPU box.

Run from the repo root on a machine with an MI355X and the prebuilt oracle/_ref/ files
(`make -C oracle ref` where the reference tree exists, then e.g.
`gpurun -- python tests/golden/make_reference_fixtures.py gpurun_out/golden`, then copy the .npz
files into tests/golden/).  The scenes come from opencl_pathtracer_amd.scenes + the product BVH
builder (itself pinned to the reference's BVH_Create by tests/test_bvh.py); the images, counts and
histograms come from oracle/_ref/ref_kernel_<case>.hsaco = Kernel/PathTracer_FullKernel.cl compiled
unmodified for gfx950.  Stored per case and iteration range: imageColor, imageRayNb, the three
histograms, plus a digest of the scene arrays so a test can tell when the generator changed.
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402
from opencl_pathtracer_amd import scenes, bvh_create  # noqa: E402


def scene_digest(sc):
    h = hashlib.sha256()
    for a in (sc.bvh["son1Id"], sc.bvh["son2Id"], sc.bvh["triangleStartIndex"], sc.bvh["nbTriangles"],
              sc.bvh["trianglesAABB"]["pMin"], sc.bvh["trianglesAABB"]["pMax"], sc.triangulation["S1"],
              sc.triangulation["S2"], sc.triangulation["S3"], sc.triangulation["N"], sc.lights["position"],
              sc.materiaux["simpleColor"], sc.texturesData):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def full_scene_digest(sc):
    """Every byte the kernel can read (the fuzzed scenes vary all of it)."""
    h = hashlib.sha256(scene_digest(sc).encode())
    for a in (sc.bvh["trianglesAABB"]["isEmpty"], sc.bvh["cutAxis"], sc.bvh["isLeaf"], sc.triangulation, sc.lights, sc.materiaux, sc.textures, sc.texturesData, sc.sky, sc.cameraPosition, sc.cameraDirection,
              sc.cameraRight, sc.cameraUp):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def fuzz_fixtures(out_dir):
    """Digests of the reference's results (both builds, 8 spp) on the seeded scenes of cases.FUZZ_FIXTURES."""
    import warnings
    digests = {}
    for name in cases.FUZZ_FIXTURES:
        case, w, h, d = cases.FUZZ_CASE[int(name.rsplit("_l", 1)[1])]
        if not (O.have_ref_kernel(case) and O.have_ref_kernel(case, strict=True)):
            print("skip", name, "(no code object)")
            continue
        sc = cases.build_fuzz(name, w, h)
        digests[name + "_scene"] = full_scene_digest(sc)
        for strict, key in ((True, name), (False, name + "_default")):
            color, count, (dep, bbx, tri), _ = O.ref_gpu_render(case, sc, w, h, d, cases.FEATURE_SPP, strict=strict)
            digests[key] = cases.result_digest(color, count, dep, bbx, tri)
        print("fuzz", name, digests[name][:16], digests[name + "_default"][:16], flush=True)
    np.savez_compressed(os.path.join(out_dir, "ref_fuzz.npz"), **{k: np.array(v) for k, v in digests.items()})


def main(out_dir, features_only=False, fuzz_only=False, only=()):
    os.makedirs(out_dir, exist_ok=True)
    if fuzz_only:
        return fuzz_fixtures(out_dir)
    for case, (name, sampler, w, h, d) in ({} if features_only else cases.CASES).items():
        if only and case not in only:
            continue
        if not O.have_ref_kernel(case):
            print("skip", case, "(no code object)")
            continue
        sc = bvh_create(scenes.build(name, w, h))
        out = {"scene_digest": np.array(scene_digest(sc))}
        for first, n in cases.FIXTURE_RANGES:
            color, count, (dep, bbx, tri), ms = O.ref_gpu_render(case, sc, w, h, d, n, first_iteration=first)
            tag = f"it{first}_{n}"
            out[tag + "_color"] = color
            out[tag + "_count"] = count
            out[tag + "_depths"] = dep
            nzb, nzt = np.flatnonzero(bbx), np.flatnonzero(tri)
            out[tag + "_bbx_idx"], out[tag + "_bbx_val"] = nzb.astype(np.uint16), bbx[nzb]
            out[tag + "_tri_idx"], out[tag + "_tri_val"] = nzt.astype(np.uint16), tri[nzt]
            print(f"{case} [{first},{first + n}): {ms:.1f} ms, depths {dep.tolist()}", flush=True)
        if O.have_ref_kernel(case, strict=True):
            # second legal build of the same source (-ffp-contract=off, correctly rounded divide/sqrt):
            # (a) the reference's distance to itself, (b) traversal-work histograms under IEEE arithmetic
            s_color, s_count, (s_dep, s_bbx, s_tri), _ = O.ref_gpu_render(case, sc, w, h, d, 16, strict=True)
            d_color = out["it0_8_color"] + out["it8_8_color"]
            out["noise_floor_rms_16spp"] = cases.rms_per_channel(s_color, s_count, d_color, s_count)
            k = np.arange(5000, dtype=np.int64)
            out["strict_16spp_work"] = np.array([(s_bbx.astype(np.int64) * k).sum(), (s_tri.astype(np.int64) * k).sum()])
            out["strict_16spp_depths"] = s_dep
            out["strict_16spp_color_mean"] = s_color[..., :3].mean(axis=(0, 1))
            # (c) the strict build's outputs themselves, iterations 0..3: the oracle and the integrator follow the same
            # arithmetic (IEEE operations + the platform library's dot / normalize / sin / cos) and must equal them bit for bit
            e_color, e_count, (e_dep, e_bbx, e_tri), _ = O.ref_gpu_render(case, sc, w, h, d, cases.STRICT_SPP, strict=True)
            out["strict_exact_color"], out["strict_exact_count"], out["strict_exact_depths"] = e_color, e_count, e_dep
            nzb, nzt = np.flatnonzero(e_bbx), np.flatnonzero(e_tri)
            out["strict_exact_bbx_idx"], out["strict_exact_bbx_val"] = nzb.astype(np.uint16), e_bbx[nzb]
            out["strict_exact_tri_idx"], out["strict_exact_tri_val"] = nzt.astype(np.uint16), e_tri[nzt]
        one, _, _, _ = O.ref_gpu_render(case, sc, w, h, d, 1)
        out["it0_1_color"] = one
        np.savez_compressed(os.path.join(out_dir, f"ref_{case}.npz"), **out)
    if only:
        return
    # one kernel feature per scene through the strict build: digests of image, counts and histograms (8 spp each);
    # `<feature>_default`: the same through the default build (what the reference's own build line produces)
    fcase, fw, fh, fd = cases.FEATURE_CASE
    if O.have_ref_kernel(fcase, strict=True):
        digests = {}
        for feature in scenes.FEATURES:
            sc = bvh_create(scenes.build("feat_" + feature, fw, fh))
            color, count, (dep, bbx, tri), _ = O.ref_gpu_render(fcase, sc, fw, fh, fd, cases.FEATURE_SPP, strict=True)
            digests[feature] = cases.result_digest(color, count, dep, bbx, tri)
            digests[feature + "_scene"] = scene_digest(sc)
            color, count, (dep, bbx, tri), _ = O.ref_gpu_render(fcase, sc, fw, fh, fd, cases.FEATURE_SPP)
            digests[feature + "_default"] = cases.result_digest(color, count, dep, bbx, tri)
            print("feature", feature, digests[feature][:16], digests[feature + "_default"][:16], flush=True)
        np.savez_compressed(os.path.join(out_dir, f"ref_{fcase}_features.npz"), **{k: np.array(v) for k, v in digests.items()})
    fuzz_fixtures(out_dir)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    # usage: make_reference_fixtures.py [out_dir [case ...]] [--features-only | --fuzz-only]   (case names: only those cases)
    main(args[0] if args else os.path.join(ROOT, "gpurun_out", "golden"), features_only="--features-only" in sys.argv,
         fuzz_only="--fuzz-only" in sys.argv, only=tuple(args[1:]))
