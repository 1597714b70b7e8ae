"""Register / LDS budget of the shipped wavefront-kernel specialisation, read from hipcc's resource remarks (cross-compiles
without a GPU).  The kernel's speed hangs on 5 waves per SIMD (96 VGPRs) with no scratch traffic inside the traversal
loop; hipcc's register allocation is fragile, so an innocent-looking edit can cost a wave of occupancy."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


# spilled registers: a regression budget per production specialisation (general shading, plain shading).  The plain-scene
# kernel holds its state in 96 registers; the general one does not and spills INSIDE THE PATH-LOGIC PASS (never in the
# traversal loop, asserted below).  Round 3 measured what those spills cost: the same kernel at 4 waves per SIMD - 128
# registers, no spill at all - is 1-4 % SLOWER on every scene (material mix 4K 2489 -> 2461 Msamples/s), so the
# budget pins today's figures (+ a little slack for compiler noise) instead of forcing the plain kernel's.
SPILL_BUDGET = {"general": 72, "plain": 8}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("arithmetic", [0, 1])
def test_wavefront_kernel_keeps_five_waves_per_simd(arithmetic):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import resources
    res = resources(ROOT, arithmetic)  # key = STATS PRE SS PLAIN NANSAFE as five 0/1 digits
    # the production specialisations: no scheduler statistics, no adaptive sampling; precomputed triangles or not, general /
    # plain shading, with / without the NaN check of scenes whose records yield NaN distances
    # ("01000b64", "01010b64": the two production kernels with workgroups of 64 lanes, launched where the tree's depth leaves a CU
    # fewer than five wide workgroups)
    assert {"01000b64", "01010b64"} <= set(res), sorted(res)
    production = {k: v for k, v in res.items() if k[0] == "0" and k[2] == "0"}
    assert {"01000", "01010", "01001", "01011", "00000", "00001", "01000b64", "01010b64"} <= set(production), sorted(res)
    for key, figures in production.items():
        assert figures["VGPRs"] <= 96 and figures["Occupancy"] >= 5, (key, figures)
        nan_safe = key[4] == "1"  # (scenes whose records yield NaN distances: a few more live values, rare scenes - a little slack)
        assert figures["VGPRs Spill"] <= SPILL_BUDGET["plain" if key[3] == "1" else "general"] + (4 if nan_safe else 0), (key, figures)
        # the traversal loop (the depth-2 blocks) must not touch scratch: spills belong to the path-logic pass
        assert figures["loop scratch"] <= (3 if nan_safe else 0), (key, figures)
    # the instrumented ones (<true, ...>: --scheduler-stats, SUPER_SAMPLING) carry more state and may reload a few words
    for key, figures in res.items():
        assert figures["VGPRs"] <= 96 and figures["Occupancy"] >= 5 and figures["loop scratch"] <= 12, (key, figures)
