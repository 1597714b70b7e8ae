"""The plumbing bench.py relies on for multi-GPU: accumulators owned by a torch tensor (ptmi_bind_accumulators),
kernels on a caller-owned stream (ptmi_set_stream), results visible to torch ops queued on that stream."""
import numpy as np
import pytest

import oracle_ffi as O

pytestmark = pytest.mark.gpu


def test_bound_accumulators_and_external_stream(scene_factory):
    import torch
    from opencl_pathtracer_amd import Backend
    from opencl_pathtracer_amd.distributed import FusedAccumulators, shard_iterations
    w, h, d, spp = 96, 96, 8, 6
    sc = scene_factory("matmix", w, h)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    stream = torch.cuda.Stream(dev)
    be = Backend().setup_context(w, h, d, sc.lightsSize)
    be.initialize_memory(sc)
    fb = FusedAccumulators(w, h, dev)
    fb.bind(be)
    be.set_stream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        # two "ranks" worth of shards rendered one after the other into the same fused buffer
        for rank in (0, 1):
            first, n = shard_iterations(0, spp, rank, 2)
            be.render(first, n)
        total = fb.buffer.sum()  # a torch op ordered after the kernels on the same stream
    stream.synchronize()
    color, count = fb.images()
    o_color, o_count, _, _ = O.oracle_render(sc, w, h, d, spp)
    assert np.array_equal(color.view(np.uint32), o_color.view(np.uint32)) and np.array_equal(count, o_count)
    assert abs(float(total) - float(o_color.astype(np.float64).sum() + o_count.sum())) <= 1e-3 * float(total)
    # a progressive-display snapshot taken on the render stream equals the accumulators (single process: no collective)
    from opencl_pathtracer_amd.distributed import ProgressiveDisplay
    shown = ProgressiveDisplay(fb)
    with torch.cuda.stream(stream):
        shown.submit()
    p_color, p_count = shown.images()
    assert np.array_equal(p_color, color) and np.array_equal(p_count, count)
    # device_accumulators reports the bound tensor; unbinding goes back to the context's own (zeroed) buffers
    a, b = be.device_accumulators()
    assert a == fb.color.data_ptr() and b == fb.count.data_ptr()
    be.bind_accumulators(0, 0)
    own, own_n = be.read_image()
    assert not own.any() and not own_n.any()
    be.release()
