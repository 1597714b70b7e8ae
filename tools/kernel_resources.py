#!/usr/bin/env python3
"""Register budget of every render_wavefront_kernel instantiation (hipcc cross-compiles without a GPU): VGPRs, spilled VGPRs,
scratch bytes, occupancy, and the number of scratch instructions INSIDE the traversal loop (depth-2 blocks), plus the static
instruction mix of that loop.  usage: tools/kernel_resources.py [source tree, default the repo] [arithmetic 0|1, default 1]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resources(tree=ROOT, arithmetic=1, extra=()):
    csrc = os.path.join(tree, "opencl_pathtracer_amd", "csrc")
    out = os.path.join(tempfile.mkdtemp(), "wf.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
           f"-DPTMI_DEFAULT_ARITHMETIC={arithmetic}", "-I" + os.path.join(tree, "include"), "-I" + csrc, "--cuda-device-only", "-S",
           os.path.join(csrc, "kernel_wavefront.hip"), "-o", out, "-Rpass-analysis=kernel-resource-usage", *extra]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr[-3000:])
    res = {}
    for block in re.split(r"Function Name: ", r.stderr)[1:]:
        m = re.match(r"\w+?23render_wavefront_kernelI((?:Lb[01]E)+)(?:Li(\d+)E)?E", block)
        if not m:
            continue
        key = "".join(re.findall(r"Lb([01])E", m.group(1))) + ("b" + m.group(2) if m.group(2) not in (None, "256") else "")  # STATS PRE SS PLAIN NANSAFE [bN: lanes per workgroup]
        res[key] = {k.strip(): int(v) for k, v in re.findall(r"remark: [^\n]*?\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", block)}
    name, depth2 = None, False
    for line in open(out):
        m = re.match(r"\w+?23render_wavefront_kernelI((?:Lb[01]E)+)(?:Li(\d+)E)?E\w*:", line)
        if m:
            name, depth2 = "".join(re.findall(r"Lb([01])E", m.group(1))) + ("b" + m.group(2) if m.group(2) not in (None, "256") else ""), False
            res[name]["loop"] = Counter()
        elif re.match(r"(\.LBB|; %bb\.)", line):
            depth2 = "Depth=2" in line
        elif depth2 and name is not None and re.match(r"\s+[a-z]", line):
            res[name]["loop"][line.split()[0]] += 1
    for v in res.values():
        c = v.pop("loop")
        tot = lambda p: sum(n for i, n in c.items() if i.startswith(p))
        v["loop scratch"] = tot("scratch_")
        v["loop VALU"], v["loop SALU"], v["loop LDS"], v["loop VMEM"] = tot("v_"), tot("s_"), tot("ds_"), tot("global_") + tot("buffer_")
    return res


if __name__ == "__main__":
    tree = sys.argv[1] if len(sys.argv) > 1 else ROOT
    arith = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    res = resources(tree, arith)
    print("STATS PRE SS PLAIN NANSAFE | VGPRs spilled scratch occupancy | traversal loop: scratch VALU SALU LDS VMEM")
    for k in sorted(res):
        v = res[k]
        print("   ".join(k[:5]) + ("  " + k[5:] if len(k) > 5 else ""), "|", v["VGPRs"], v["VGPRs Spill"], v["ScratchSize"], v["Occupancy"], "|", v["loop scratch"], v["loop VALU"], v["loop SALU"],
              v["loop LDS"], v["loop VMEM"])
