python -m pytest tests -m gpu -q -x > gpurun_out/r02_t18_pytest.log 2>&1; tail -3 gpurun_out/r02_t18_pytest.log
python bench.py > gpurun_out/r02_final_bench.json 2> gpurun_out/r02_final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/r02_final_bench.json')); print(d['value'], d['Mpaths/s']); print(json.dumps(d['roofline'].get('binding'))[:600]); print(json.dumps(d['boundary'])[:900])"
