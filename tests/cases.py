"""The parity cases shared by the GPU tests, the fixture generator and the CPU golden tests.

name -> (scene, sampler, W, H, depth).  Names equal the -D specialisations of the reference kernel
listed in oracle/ref_configs.txt, so each case can also be run through the reference itself.
"""
from opencl_pathtracer_amd import structs as S

CASES = {
    "cornell_64x48_d4": ("cornell", S.JITTERED, 64, 48, 4),
    "cornell_128x128_d8": ("cornell", S.JITTERED, 128, 128, 8),
    "cornell_64x48_d4_uni": ("cornell", S.UNIFORM, 64, 48, 4),
    "matmix_96x96_d8": ("matmix", S.JITTERED, 96, 96, 8),
    "matmix_96x96_d8_uni": ("matmix", S.UNIFORM, 96, 96, 8),
    "tris20k_96x64_d6": ("tris20k", S.JITTERED, 96, 64, 6),
    "tris1m_160x90_d10": ("tris1m", S.JITTERED, 160, 90, 10),
    # scenes.maya_like at the CPU checker's size: the records of the configs[4] stand-in (1024x1024 file textures, 512x512 sky faces)
    "mayalike_s_96x96_d8": ("mayalike_s", S.JITTERED, 96, 96, 8),
}
SMALL = [k for k in CASES if not k.startswith("tris1m")]
FIXTURE_RANGES = [(0, 8), (8, 8)]  # (first iteration, count): two shards of a 16-spp render
STRICT_SPP = 4                      # iterations 0..3 of the reference's strict build are stored in full (bit-exact pin)
FEATURE_CASE = ("feat_64x64_d8", 64, 64, 8)  # one specialisation of the reference kernel for every scenes.feature_scene
FEATURE_SPP = 8
# scenes.fuzz_scene / corrupt_records scenes whose reference results (both builds) are committed as digests
# (tests/golden/ref_fuzz.npz): h = hostile records (NaN distances), r = records no importer writes, _l<n> = lights.
# 1 light -> code object feat_64x64_d8, 3 lights -> matmix_96x96_d8
# t = scenes.corrupt_tree on the built tree
FUZZ_FIXTURES = ["fuzz0_l1", "fuzz0h_l1", "fuzz1_l3", "fuzz1h_l3", "fuzz2_l1", "fuzz3_l1", "fuzz5h_l1", "fuzz7h_l3", "fuzz40r_l1", "fuzz41hr_l1",
                 "fuzz42r_l3", "fuzz43hr_l3", "fuzz46r_l1", "fuzz47hr_l3", "fuzz60t_l1", "fuzz61rt_l3", "fuzz63ht_l1", "fuzz67rt_l1"]
FUZZ_CASE = {1: ("feat_64x64_d8", 64, 64, 8), 3: ("matmix_96x96_d8", 96, 96, 8)}


def build_fuzz(name, w, h):
    """A FUZZ_FIXTURES scene with its tree: scenes.build + the product's BVH builder (+ scenes.corrupt_tree for a `t` name)."""
    import re
    import warnings
    from opencl_pathtracer_amd import scenes, bvh_create
    m = re.fullmatch(r"fuzz(\d+)(h?)(r?)(t?)_l(\d+)", name)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # (the hostile scenes divide 0 by 0 on purpose, as the importer would)
        sc = bvh_create(scenes.build(f"fuzz{m.group(1)}{m.group(2)}{m.group(3)}_l{m.group(5)}", w, h))
        if m.group(4):
            scenes.corrupt_tree(sc, int(m.group(1)))
    return sc


def result_digest(color, count, depths, bbx, tri):
    """SHA-256 over the bits of a render: image, sample counts and the three histograms."""
    import hashlib
    import numpy as np
    h = hashlib.sha256()
    for a, t in ((color, np.float32), (count, np.float32), (depths, np.uint32), (bbx, np.uint32), (tri, np.uint32)):
        h.update(np.ascontiguousarray(a, dtype=t).tobytes())
    return h.hexdigest()


def rms_per_channel(a_color, a_count, b_color, b_count):
    """Per-channel RMS of the displayed images sum/n (the north-star parity metric)."""
    import numpy as np
    ia = a_color[..., :3] / np.maximum(a_count, 1)[..., None]
    ib = b_color[..., :3] / np.maximum(b_count, 1)[..., None]
    return np.sqrt(((ia.astype(np.float64) - ib.astype(np.float64)) ** 2).mean(axis=(0, 1)))


def diff_report(a_color, a_count, b_color, b_count, flip_threshold=1e-3):
    """Separates rare 'flipped' pixels (a sample took another branch: chaotic, last-bit arithmetic) from the
    systematic rounding difference.  Returns dict(rms, n_flipped, rms_without_flipped, max_abs, max_value)."""
    import numpy as np
    ia = (a_color[..., :3] / np.maximum(a_count, 1)[..., None]).astype(np.float64)
    ib = (b_color[..., :3] / np.maximum(b_count, 1)[..., None]).astype(np.float64)
    d = np.abs(ia - ib).max(axis=-1)
    flipped = d > flip_threshold
    rest = np.where(flipped[..., None], 0.0, ia - ib)
    return {"rms": np.sqrt(((ia - ib) ** 2).mean(axis=(0, 1))).max(), "n_flipped": int(flipped.sum()),
            "rms_without_flipped": float(np.sqrt((rest ** 2).mean(axis=(0, 1))).max()), "max_abs": float(d.max()),
            "max_value": float(max(ia.max(), ib.max())), "n_pixels": int(d.size)}


FLIP_REL = 1e-3          # a sample whose radiance moved by more than this (relative) took another branch
MAX_FLIP_FRACTION = 0.01  # measured: <= 0.5 % of samples, all in chaotic zones (grazing shadow rays, edges)


def sample_agreement(a_color, b_color):
    """Per-sample statistics of two 1-spp renders (one path per pixel): dict(flip_fraction, median_rel, bias).
    bias = signed mean difference of the flipped samples relative to the mean radiance: chaotic flips are
    unbiased, a semantic difference is not."""
    import numpy as np
    a, b = a_color[..., :3].astype(np.float64), b_color[..., :3].astype(np.float64)
    scale = np.maximum(np.maximum(np.abs(a), np.abs(b)).max(-1), 1e-12)
    rel = np.abs(a - b).max(-1) / scale
    flipped = rel > FLIP_REL
    return {"flip_fraction": float(flipped.mean()), "median_rel": float(np.median(rel)),
            "bias": float((a - b).mean() / max(a.mean(), 1e-12)), "n": int(rel.size)}
