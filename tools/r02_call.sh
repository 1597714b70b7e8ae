python -m pytest tests -m gpu -q -x > gpurun_out/r02_t16_pytest.log 2>&1; tail -3 gpurun_out/r02_t16_pytest.log
for i in 1 2; do
python bench.py --scene cornell --depth 8 --steps 8 --warmup 2 --no-cpu-baseline --no-boundary > gpurun_out/r02_bench_e_cornell.json 2>>gpurun_out/r02_bench_e.err; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_e_cornell.json')); print('cornell', d['value'], d['Mpaths/s'])"
python bench.py --scene matmix --width 3840 --height 2160 --depth 16 --steps 3 --warmup 1 --no-cpu-baseline --no-boundary > gpurun_out/r02_bench_e_matmix.json 2>>gpurun_out/r02_bench_e.err; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_e_matmix.json')); print('matmix4k', d['value'], d['Mpaths/s'])"
done
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-boundary > gpurun_out/r02_bench_e.json 2>>gpurun_out/r02_bench_e.err; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_e.json')); print('tris1m', d['value'], d['Mpaths/s'])"
STEPS=4 BENCH_ARGS="--spp-per-step 32" bash tools/run_variants.sh it32
