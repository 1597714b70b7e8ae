"""Samples-per-pixel sharding across GPUs.

A sample is a pure function of (pixel, iteration id) -- the seed is derived from them alone
(Kernel/PathTracer_FullKernel_header.cl:255-264) -- the scene is read-only and the accumulators are
plain sums (FullKernel.cl:1339-1345).  So the iteration ids of a render are PARTITIONED over ranks,
every rank renders the full image for its ids on a full scene replica with no data-path exchange,
and ONE collective at the end sums the accumulators onto rank 0: a reduce of a fused
float[5*W*H] buffer (imageColor float4 + imageRayNb float; 41.5 MB at 1080p) over RCCL/xGMI.
The reference has no multi-device path at all (single context, devices[0], PathTracer_OpenCL.cpp:363-366).

torch is plumbing here: device memory for the fused buffer, the process group, the collective.
"""
import torch
import torch.distributed as dist


def shard_iterations(first_iteration, n_iterations, rank, world_size):
    """Contiguous block partition of [first, first+n) -> (first_r, n_r); blocks differ by at most one id."""
    base, extra = divmod(n_iterations, world_size)
    n_r = base + (1 if rank < extra else 0)
    first_r = first_iteration + rank * base + min(rank, extra)
    return first_r, n_r


class FusedAccumulators:
    """imageColor and imageRayNb as two views of ONE tensor so a single collective moves both."""

    def __init__(self, width, height, device):
        self.npix = width * height
        self.width, self.height = width, height
        self.buffer = torch.zeros(5 * self.npix, dtype=torch.float32, device=device)
        self.color = self.buffer[: 4 * self.npix]
        self.count = self.buffer[4 * self.npix:]

    def bind(self, backend):
        """Make the integrator accumulate straight into this tensor (ptmi_bind_accumulators)."""
        backend.bind_accumulators(self.color.data_ptr(), self.count.data_ptr())

    def reduce_to(self, dst=0, group=None):
        """The one collective of a sharded render: sum over ranks onto `dst`."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.reduce(self.buffer, dst=dst, op=dist.ReduceOp.SUM, group=group)

    def images(self):
        c = self.color.view(self.height, self.width, 4).cpu().numpy()
        n = self.count.view(self.height, self.width).cpu().numpy()
        return c, n


def reduce_statistics(depths, bbx, tri, dst=0, group=None, device=None):
    """Histograms are integer sums too; int64 on the wire (uint32 is not a collective dtype)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return depths, bbx, tri
    import numpy as np
    t = torch.from_numpy(np.concatenate([depths, bbx, tri]).astype(np.int64))
    if device is not None:
        t = t.to(device)
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM, group=group)
    a = t.cpu().numpy()
    nd, nb = len(depths), len(bbx)
    return a[:nd].astype(np.uint64), a[nd:nd + nb].astype(np.uint64), a[nd + nb:].astype(np.uint64)
