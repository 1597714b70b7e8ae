"""The N>1 path on CPU: world_size-2 gloo.  The product's shard + fused-reduce code
(opencl_pathtracer_amd.distributed) is exercised with the CPU oracle standing in for the per-rank
renderer (no GPU here); the GPU box runs the same code over RCCL in bench.py."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, D, SPP = 48, 32, 4, 10


def test_shard_iterations_partition():
    from opencl_pathtracer_amd.distributed import shard_iterations
    for first, n, world in [(0, 10, 2), (0, 4096, 8), (5, 7, 3), (0, 1, 4), (100, 0, 2)]:
        ids = []
        for r in range(world):
            f, k = shard_iterations(first, n, r, world)
            ids += list(range(f, f + k))
        assert ids == list(range(first, first + n))
        sizes = [shard_iterations(first, n, r, world)[1] for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, out_path):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_ffi as O
    from opencl_pathtracer_amd import scenes, bvh_create
    from opencl_pathtracer_amd.distributed import FusedAccumulators, shard_iterations, reduce_statistics
    sc = bvh_create(scenes.cornell_box(W, H))
    first, n = shard_iterations(0, SPP, rank, world)
    color, count, (dep, bbx, tri), _ = O.oracle_render(sc, W, H, D, n, first_iteration=first, n_threads=2)
    fb = FusedAccumulators(W, H, torch.device("cpu"))
    fb.color.copy_(torch.from_numpy(color.reshape(-1)))
    fb.count.copy_(torch.from_numpy(count.reshape(-1)))
    fb.reduce_to(0)
    dep, bbx, tri = reduce_statistics(dep, bbx, tri, dst=0)
    if rank == 0:
        c, k = fb.images()
        np.savez(out_path, color=c, count=k, depths=dep)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_spp_shards_reduce_to_the_single_rank_image(built, tmp_path):
    import oracle_ffi as O
    from opencl_pathtracer_amd import scenes, bvh_create
    out = str(tmp_path / "rank0.npz")
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    sc = bvh_create(scenes.cornell_box(W, H))
    color, count, (dep, _, _), _ = O.oracle_render(sc, W, H, D, SPP)
    assert np.array_equal(got["count"], count) and np.array_equal(got["depths"], dep)
    # same samples, only the fp32 summation order differs: (r0+..+r4) + (r5+..+r9) vs sequential
    assert np.allclose(got["color"], color, rtol=2e-6, atol=1e-6)
