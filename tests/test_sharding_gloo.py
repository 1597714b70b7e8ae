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
    # (the arithmetic bench.py renders in by default: the reference's own build's)
    color, count, (dep, bbx, tri), _ = O.oracle_render(sc, W, H, D, n, first_iteration=first, n_threads=2, default_arithmetic=True)
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
    color, count, (dep, _, _), _ = O.oracle_render(sc, W, H, D, SPP, default_arithmetic=True)
    assert np.array_equal(got["count"], count) and np.array_equal(got["depths"], dep)
    # same samples, only the fp32 summation order differs: (r0+..+r4) + (r5+..+r9) vs sequential
    assert np.allclose(got["color"], color, rtol=2e-6, atol=1e-6)


# ---- SUPER_SAMPLING: the variance image of two shards ------------------------------------------------------------

def _welford(samples):
    """(sum, n, M2) exactly as the kernel builds them, one sample at a time in float32 (FullKernel.cl:1339-1349)."""
    f = np.float32
    s = np.zeros(samples.shape[1:], f)
    m2 = np.zeros(samples.shape[1:], f)
    n = f(0)
    for k, x in enumerate(samples.astype(f)):
        after = s + x
        n_after = f(n + 1)
        if k != 0:
            m2 = m2 + (x - s / n) * (x - after / n_after)
        s, n = after, n_after
    return s, n, m2


def test_merge_moments_equals_single_pass():
    from opencl_pathtracer_amd.distributed import merge_moments
    rs = np.random.RandomState(4)
    P = 500
    x = rs.gamma(2.0, 0.3, (40, P, 4)).astype(np.float32)
    cut = rs.randint(0, 41, P)  # pixel p: samples [0, cut) on shard a, the rest on shard b (0 and 40 = one empty shard)
    sa = np.zeros((P, 4), np.float32); na = np.zeros(P, np.float32); ma = np.zeros((P, 4), np.float32)
    sb = np.zeros((P, 4), np.float32); nb = np.zeros(P, np.float32); mb = np.zeros((P, 4), np.float32)
    want_s = np.zeros((P, 4)); want_m = np.zeros((P, 4))
    for p in range(P):
        if cut[p] > 0:
            sa[p], na[p], ma[p] = _welford(x[:cut[p], p])
        if cut[p] < 40:
            sb[p], nb[p], mb[p] = _welford(x[cut[p]:, p])
        xs = x[:, p].astype(np.float64)
        want_s[p] = xs.sum(0)
        want_m[p] = ((xs - xs.mean(0)) ** 2).sum(0)
    t = torch.from_numpy
    s, n, m2 = merge_moments(t(sa), t(na), t(ma), t(sb), t(nb), t(mb))
    assert np.array_equal(n.numpy(), np.full(P, 40, np.float32))
    assert np.allclose(s.numpy(), want_s, rtol=1e-5)
    assert np.allclose(m2.numpy(), want_m, rtol=2e-4, atol=1e-5)
    assert np.isfinite(m2.numpy()).all()


def _ss_worker(rank, world, port, out_path):
    sys.path[:0] = [ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opencl_pathtracer_amd.distributed import reduce_super_sampling
    rs = np.random.RandomState(9)
    x = rs.gamma(2.0, 0.3, (24, 64, 4)).astype(np.float32)
    mine = x[rank::world]  # interleaved shards
    s = np.zeros((64, 4), np.float32); m = np.zeros((64, 4), np.float32)
    for p in range(64):
        s[p], _, m[p] = _welford(mine[:, p])
    n = np.full(64, len(mine), np.float32)
    cs, cn, cm = reduce_super_sampling(torch.from_numpy(s), torch.from_numpy(n), torch.from_numpy(m), dst=0)
    if rank == 0:
        np.savez(out_path, s=cs.numpy(), n=cn.numpy(), m=cm.numpy(), x=x)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_super_sampling_variance_merge(tmp_path):
    out = str(tmp_path / "ss.npz")
    port = 31500 + os.getpid() % 2000
    mp.spawn(_ss_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    xs = got["x"].astype(np.float64)
    assert np.array_equal(got["n"], np.full(64, 24, np.float32))
    assert np.allclose(got["s"], xs.sum(0), rtol=1e-5)
    assert np.allclose(got["m"], ((xs - xs.mean(0)) ** 2).sum(0), rtol=2e-4, atol=1e-5)


def _progressive_worker(rank, world, port, out_path):
    sys.path[:0] = [ROOT]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from opencl_pathtracer_amd.distributed import FusedAccumulators, ProgressiveDisplay
    fb = FusedAccumulators(8, 4, torch.device("cpu"))
    shown = ProgressiveDisplay(fb, dst=0)
    pictures = []
    for step in range(3):  # "render" = every rank adds (rank + 1) to every accumulator entry
        fb.buffer += float(rank + 1)
        shown.submit()
        c, n = shown.images()
        pictures.append((float(c[0, 0, 0]), float(n[0, 0])))
    own = float(fb.buffer[0])  # the accumulators themselves were never reduced
    fb.reduce_to(0)
    if rank == 0:
        np.savez(out_path, pictures=np.array(pictures), own=own, final=fb.buffer.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_progressive_display_reduces_snapshots_not_accumulators(tmp_path):
    out = str(tmp_path / "prog.npz")
    port = 33500 + os.getpid() % 2000
    mp.spawn(_progressive_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got["pictures"].tolist() == [[3.0, 3.0], [6.0, 6.0], [9.0, 9.0]]  # (1 + 2) per step, summed over both ranks
    assert got["own"] == 3.0 and (got["final"] == 9.0).all()
