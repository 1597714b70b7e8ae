"""Runs in a subprocess with libasan / libubsan preloaded (tests/test_oracle_golden.py::test_oracle_under_sanitizers): the CPU
oracle compiled with -fsanitize=address,undefined,float-cast-overflow on the committed fuzz scenes (hostile records, NaN
distances, corrupted records and trees) - no access outside the scene's arrays, no undefined conversion, in either arithmetic.
usage: oracle_asan.py STRICT_SO DEFAULT_ARITHMETIC_SO"""
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
warnings.simplefilter("ignore")
import cases  # noqa: E402
import oracle_ffi as O  # noqa: E402

O.ORACLE_LIB, O.ORACLE_DA_LIB = sys.argv[1], sys.argv[2]
n = 0
for name in cases.FUZZ_FIXTURES:
    case, w, h, d = cases.FUZZ_CASE[int(name.rsplit("_l", 1)[1])]
    sc = cases.build_fuzz(name, w, h)
    for da in (False, True):
        O.oracle_render(sc, w, h, d, 2, default_arithmetic=da, n_threads=2)
        n += 1
print("clean", n)
