#!/bin/bash
# First contact with a node that has MORE THAN ONE GPU (none of this has run on one: DESIGN.md 6).  Run from the repo root on
# the node, after `python __graft_entry__.py` has built the libraries.  Every step prints what it proves; stop at the first
# failure - later steps assume the earlier ones.  Never more than $N ranks touch the cards (default: all visible devices).
set -e
N=${N:-$(python -c "import torch; print(torch.cuda.device_count())")}
[ "$N" -ge 2 ] || { echo "this script needs >= 2 GPUs (found $N)"; exit 2; }
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
echo "== 1. in-library form, peer copies (default): $N devices listed in ptmi_config.devices"
python - <<PY
import numpy as np, opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend
w, h, d, spp, n = 256, 192, 6, 12, $N
sc = pt.bvh_create(pt.scenes.build("tris20k", w, h))
one = pt.render_scene(sc, w, h, d, spp, flags=backend.FLAG_DEFAULT_ARITHMETIC)
many = pt.render_scene(sc, w, h, d, spp, flags=backend.FLAG_DEFAULT_ARITHMETIC, devices=list(range(n)))
assert np.array_equal(one[1], many[1]) and one[3] == many[3] and all(np.array_equal(a, b) for a, b in zip(one[2], many[2])), "counts / counters / histograms differ"
assert np.allclose(one[0], many[0], rtol=2e-6, atol=1e-6), "image differs by more than the order of the float additions"
print("   ok: counts, counters and histograms exact, image equal up to summation order on", n, "devices")
PY
echo "== 2. in-library form, PTMI_REDUCE=rccl: the same render through ncclReduce (communicators from ncclCommInitAll)"
PTMI_REDUCE=rccl python - <<PY
import numpy as np, opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend
w, h, d, spp, n = 256, 192, 6, 12, $N
sc = pt.bvh_create(pt.scenes.build("tris20k", w, h))
import os
os.environ.pop("PTMI_REDUCE"); peer = pt.render_scene(sc, w, h, d, spp, flags=backend.FLAG_DEFAULT_ARITHMETIC, devices=list(range(n)))
os.environ["PTMI_REDUCE"] = "rccl"
be = pt.Backend().setup_context(w, h, d, sc.lightsSize, devices=list(range(n)), flags=backend.FLAG_DEFAULT_ARITHMETIC)
be.initialize_memory(sc); be.render(0, spp); color, count = be.read_image(); path = be.reduce_path(); be.release()
print("   reduce path:", path)
assert path["rccl_state"] == 1 and path["communicators"] == n, "the collective was not used (library absent, refused, or a run-time error: see ptmi_last_error)"
assert np.array_equal(count, peer[1])
if n == 2:
    assert np.array_equal(color.view(np.uint32), peer[0].view(np.uint32)), "two devices: one addition per pixel, must be bit-equal to the peer path"
assert np.allclose(color, peer[0], rtol=2e-6, atol=1e-6)
print("   ok: ncclReduce over", n, "communicators equals the peer-copy sum", "(bit for bit)" if n == 2 else "(up to summation order)")
PY
echo "== 3. the reference's own PathTracer_Main over the shim on all devices (PTMI_DEVICES=all is the default)"
if [ -x oracle/_ref/ref_main_driver ]; then PTMI_LOG=1 oracle/_ref/ref_main_driver 2>&1 | tail -5; else echo "   (oracle/_ref/ref_main_driver not built here: skipped)"; fi
echo "== 3b. a caller that asks for one image per call and waits (the reference's loop) on all devices: each device renders ahead of its own calls"
echo "   (expect about N times the one-device figure with launches ahead, the one-device figure without)"
BLOCKING_RATE_DEVICES=$(seq -s, 0 $((N - 1))) python tools/blocking_rate.py tris1m
python tools/blocking_rate.py tris1m
echo "== 4. one process per GPU over RCCL: bench.py --gpus k, weak scaling, k = 1, 2, 4, ... <= $N"
for k in 1 2 4 8; do
  [ $k -le $N ] || break
  if [ $k -eq 1 ]; then python bench.py --gpus 1 --steps 4 --warmup 1 --no-cpu-baseline --no-boundary --no-reference-kernel > gpurun_out/multi_gpu_bench_1.json
  else python -m torch.distributed.run --nnodes=1 --nproc-per-node $k --master-addr 127.0.0.1 --master-port $((29500 + k)) bench.py --gpus $k --steps 4 --warmup 1 > gpurun_out/multi_gpu_bench_$k.json; fi
  python -c "import json; d=json.load(open('gpurun_out/multi_gpu_bench_$k.json')); print('   N=$k', round(d['value'],1), d['unit'], 'collective', d['config']['collective'])"
done
echo "== 5. BASELINE configs[3]: 4096 spp split over the ranks (strong scaling), the largest k above"
echo "   python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 bench.py --gpus $N --total-spp 4096"
echo "== 6. byte budget of DESIGN.md 6 against a copy trace (one rocprofv3 run, memory-copy trace only, no counters)"
echo "   rocprofv3 --memory-copy-trace --stats -d gpurun_out/copytrace -- python3 tools/north_star_full_size.py   (adapt: PTMI_DEVICES=all)"
