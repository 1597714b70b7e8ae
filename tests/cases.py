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
}
SMALL = [k for k in CASES if not k.startswith("tris1m")]
FIXTURE_RANGES = [(0, 8), (8, 8)]  # (first iteration, count): two shards of a 16-spp render


def rms_per_channel(a_color, a_count, b_color, b_count):
    """Per-channel RMS of the displayed images sum/n (the north-star parity metric)."""
    import numpy as np
    ia = a_color[..., :3] / np.maximum(a_count, 1)[..., None]
    ib = b_color[..., :3] / np.maximum(b_count, 1)[..., None]
    return np.sqrt(((ia.astype(np.float64) - ib.astype(np.float64)) ** 2).mean(axis=(0, 1)))
