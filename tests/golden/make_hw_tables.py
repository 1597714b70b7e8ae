"""tests/golden/rcp_gfx950.npz, sqrt_gfx950.npz (and rsq_gfx950.npz again): how the MI355X's v_rcp_f32 / v_sqrt_f32 /
v_rsq_f32 deviate from the correctly rounded functions.  The reference's DEFAULT OpenCL build divides through v_rcp_f32 of
the divisor's mantissa and takes square roots with v_sqrt_f32, so the CPU oracle's default-arithmetic build
(oracle/build/libpt_oracle_da.so) needs them the way its strict build needs v_rsq_f32 for normalize().

Made on the GPU box:   tools/microbench/hw_survey rcp  gpurun_out/hw/rcp_dev.bin      (hipcc --offload-arch=gfx950 -O2
                       tools/microbench/hw_survey sqrt gpurun_out/hw/sqrt_dev.bin       tools/microbench/hw_survey.hip)
then here:             python tests/golden/make_hw_tables.py rcp gpurun_out/hw/rcp_dev.bin   (same for sqrt, rsq)

Stored: `packed` uint8, four entries per byte, entry i in bits 2*(i%4).. of byte i/4: 0 = hardware result one ulp below the
reference formula, 1 = equal, 2 = one ulp above.
  rcp:        2^23 entries, index = mantissa of the input in [0.5, 1);  formula (float)(1.0 / (double)x)
  sqrt, rsq:  2^24 entries, index = (exponent parity << 23) | mantissa, input in [1, 2) / [2, 4);
              formulas (float)sqrt((double)x), (float)(1.0 / sqrt((double)x))
The oracle adds the deviation to the same formula evaluated on the host.
"""
import os
import sys

import numpy as np

kind, src = sys.argv[1], sys.argv[2]
assert kind in ("rcp", "sqrt", "rsq")
dev = np.fromfile(src, dtype=np.int8)
assert dev.size == (1 << 23 if kind == "rcp" else 1 << 24) and dev.min() >= -1 and dev.max() <= 1
u = (dev + 1).astype(np.uint8)
packed = (u[0::4] | (u[1::4] << 2) | (u[2::4] << 4) | (u[3::4] << 6)).astype(np.uint8)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), f"{kind}_gfx950.npz")
counts = np.bincount(u, minlength=3)
if kind == "rsq" and os.path.exists(out):
    assert np.array_equal(np.load(out)["packed"], packed), "differs from the committed v_rsq_f32 table"
    print("rsq: equal to the committed table")
else:
    np.savez_compressed(out, packed=packed, counts=counts)
print(out, os.path.getsize(out), "bytes; below / equal / above:", counts.tolist())
