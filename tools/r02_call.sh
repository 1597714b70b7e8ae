python -m pytest tests -m gpu -q -x > gpurun_out/r02_t6_pytest.log 2>&1; tail -25 gpurun_out/r02_t6_pytest.log
