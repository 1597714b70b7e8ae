"""tests/golden/rsq_gfx950.npz: how the hardware reciprocal square root of the MI355X (v_rsq_f32 - what the ROCm OpenCL
library's normalize() multiplies by, hence what the reference kernel computes on this platform) deviates from the
correctly rounded 1/sqrt(x), for every mantissa at both exponent parities (2^24 inputs).

Made on the GPU box:   tools/microbench/rsq_survey gpurun_out/rsq_dev.bin        (hipcc --offload-arch=gfx950 -O2
                                                                                   tools/microbench/rsq_survey.hip)
then here:             python tests/golden/make_rsq_table.py gpurun_out/rsq_dev.bin

Stored: `packed` uint8[2^22], four entries per byte, entry i in bits 2*(i%4).. of byte i/4: 0 = hardware result one ulp
below the correctly rounded value, 1 = equal, 2 = one ulp above.  Index = (exponent parity << 23) | mantissa.
The CPU oracle (oracle/pt_oracle.c: pto_hardware_rsq) adds the deviation to a correctly rounded 1/sqrt computed in double.
"""
import os
import sys

import numpy as np

src = sys.argv[1]
dev = np.fromfile(src, dtype=np.int8)
assert dev.size == 1 << 24 and dev.min() >= -1 and dev.max() <= 1
u = (dev + 1).astype(np.uint8)
packed = (u[0::4] | (u[1::4] << 2) | (u[2::4] << 4) | (u[3::4] << 6)).astype(np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rsq_gfx950.npz")
counts = np.bincount(u, minlength=3)
np.savez_compressed(out, packed=packed, counts=counts)
print(out, os.path.getsize(out), "bytes; below / equal / above:", counts.tolist())
