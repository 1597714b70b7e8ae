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
def test_wavefront_kernel_keeps_five_waves_per_simd(arithmetic, tmp_path):
    csrc = os.path.join(ROOT, "opencl_pathtracer_amd", "csrc")
    ns = "_ZN11ptmi_dev_da" if arithmetic else "_ZN8ptmi_dev"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           f"-DPTMI_DEFAULT_ARITHMETIC={arithmetic}",
           "-I" + os.path.join(ROOT, "include"), "-I" + csrc, "--cuda-device-only", "-S",
           os.path.join(csrc, "kernel_wavefront.hip"), "-o", str(tmp_path / "wf.s"), "-Rpass-analysis=kernel-resource-usage"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    # remarks come in blocks: "Function Name: <mangled>" followed by the figures of that function
    blocks = re.split(r"Function Name: ", r.stderr)[1:]
    # the two production specialisations: precomputed triangles, general / plain shading
    main = [b for b in blocks if b.startswith(ns + "23render_wavefront_kernelILb0ELb1ELb0ELb")]
    assert len(main) == 2, [b[:60] for b in blocks]
    for block in main:
        figures = {k: int(v) for k, v in re.findall(r"remark: [^\n]*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", block)}
        assert figures["VGPRs"] <= 96, figures
        assert figures["Occupancy"] >= 5, figures
        kind = "plain" if "ILb0ELb1ELb0ELb1E" in block[:70] else "general"
        assert figures["VGPRs Spill"] <= SPILL_BUDGET[kind], (kind, figures)
    # the traversal loop (the depth-2 loops of every specialisation) must not touch scratch in the production specialisations
    # (no scheduler statistics, no adaptive sampling): spills belong to the path-logic pass.  The instrumented ones
    # (<true, ...>: --scheduler-stats, SUPER_SAMPLING) carry more state and may reload a few words.
    depth2, name, hot_scratch = False, None, {}
    for line in open(tmp_path / "wf.s"):
        m = re.match(r"(" + ns + r"23render_wavefront_kernelILb[01]ELb[01]ELb[01]ELb[01]E)\w*:", line)
        if m:
            name, depth2 = m.group(1), False
        elif re.match(r"(\.LBB|; %bb\.)", line):
            depth2 = "Depth=2" in line
        elif depth2 and "scratch_" in line and name is not None:
            hot_scratch[name] = hot_scratch.get(name, 0) + 1
    production = {k: v for k, v in hot_scratch.items() if "ILb0E" in k}
    assert not production, f"scratch instructions inside a traversal loop: {production}"
    assert all(v <= 6 for v in hot_scratch.values()), hot_scratch  # (instrumented builds only: statistics + invariant checks, SUPER_SAMPLING)
