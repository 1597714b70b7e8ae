import sys, os, warnings
sys.path.insert(0, os.getcwd())
import opencl_pathtracer_amd as pt
from opencl_pathtracer_amd import backend, scenes, structs as S
w, h, d, spp = 96, 64, 10, 2048
for name in ("fuzz47r_l1", "fuzz40r_l1", "fuzz46r_l1", "fuzz44r_l1", "fuzz48r_l1", "fuzz50r_l1"):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        sc = pt.bvh_create(scenes.build(name, w, h))
    for sampler in (S.JITTERED, S.RANDOM):
        be = pt.Backend().setup_context(w, h, d, 1, sampler, flags=backend.FLAG_DEFAULT_ARITHMETIC)
        be.initialize_memory(sc)
        why = be.literal_kernel_reason()
        be.render(0, spp); be.synchronize()
        print(name, "sampler", sampler, "retraced", be.scheduler_stats()["paths_retraced"], "literal" if why else "", flush=True)
        be.release()
