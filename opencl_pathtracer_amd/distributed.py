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
        self._backend = None

    def bind(self, backend):
        """Make the integrator accumulate straight into this tensor (ptmi_bind_accumulators).  The zero fill above ran
        on torch's current stream and the integrator has a stream of its own: wait for the fill before binding."""
        if self.buffer.is_cuda:
            torch.cuda.current_stream(self.buffer.device).synchronize()
        backend.bind_accumulators(self.color.data_ptr(), self.count.data_ptr())
        self._backend = backend

    def reduce_to(self, dst=0, group=None, ordered=False):
        """The one collective of a sharded render: sum over ranks onto `dst`.

        Stream contract: the collective runs on torch's CURRENT stream, the launches on the backend's stream.  Unless the
        caller states that they are the same stream (``ordered=True``, after ``backend.set_stream(current_stream)`` with a
        non-default stream, as bench.py does) the host waits for the integrator here, so the reduce can never read
        accumulators that are still being written."""
        if self._backend is not None and not ordered:
            self._backend.synchronize()
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


# ---- SUPER_SAMPLING across ranks ---------------------------------------------------------------------------------
#
# With -D SUPER_SAMPLING the kernel keeps, per pixel and channel, the count n, the sum S and imageV = the sum of
# squared deviations M2, updated one sample at a time as M2 += (x - mean_before) * (x - mean_after)
# (FullKernel.cl:1346-1349, Welford).  S and n of two shards add; M2 does not:
#       M2 = M2_a + M2_b + (mean_b - mean_a)^2 * n_a * n_b / (n_a + n_b)          (Chan, Golub, LeVeque 1979)
# A pixel a shard never sampled (n = 0) contributes nothing.  The reference has no multi-device path; this is what
# makes the variance image of a sharded render the one a single device would have built from the same samples
# (up to fp32 rounding) - the per-shard stop decisions themselves are each shard's own.

def merge_moments(sum_a, n_a, m2_a, sum_b, n_b, m2_b):
    """Combine (sum, n, M2) of two disjoint sample sets; tensors of shape [..., C], [...], [..., C] (torch)."""
    n = n_a + n_b
    na, nb, nn = n_a.unsqueeze(-1), n_b.unsqueeze(-1), n.unsqueeze(-1)
    safe = lambda x: torch.where(x > 0, x, torch.ones_like(x))
    delta = sum_b / safe(nb) - sum_a / safe(na)
    cross = delta * delta * (na * nb / safe(nn))
    both = (na > 0) & (nb > 0)
    m2 = m2_a + m2_b + torch.where(both, cross, torch.zeros_like(cross))
    return sum_a + sum_b, n, m2


def reduce_super_sampling(color, count, variance, dst=0, group=None):
    """All ranks pass their (imageColor [P,4], imageRayNb [P], imageV [P,4]); rank `dst` gets the merged triple
    (the others get their own back).  One all_gather of a fused [9*P] buffer, then a pairwise merge in rank order."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return color, count, variance
    world = dist.get_world_size(group)
    p = count.numel()
    mine = torch.cat([color.reshape(-1), count.reshape(-1), variance.reshape(-1)]).contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine, group=group)
    if dist.get_rank(group) != dst:
        return color, count, variance
    split = lambda t: (t[:4 * p].view(p, 4), t[4 * p:5 * p], t[5 * p:].view(p, 4))
    s, n, m2 = split(parts[0])
    for t in parts[1:]:
        sb, nb, mb = split(t)
        s, n, m2 = merge_moments(s, n, m2, sb, nb, mb)
    return s.reshape(color.shape), n.reshape(count.shape), m2.reshape(variance.shape)


# ---- progressive display of a sharded render ---------------------------------------------------------------------
#
# The reference shows the image after every iteration (its callback runs between launches, OpenCL.cpp:97-103).  With
# the iterations spread over ranks the picture so far is the sum of everybody's partial accumulators: every K
# iterations each rank snapshots its fused buffer (a device-to-device copy on the render stream) and the snapshots are
# reduced onto `dst` on a SIDE stream, so the integrator keeps running underneath and the accumulators themselves are
# never touched by a collective before the final reduce.

class ProgressiveDisplay:
    def __init__(self, accumulators, dst=0, group=None):
        self.acc, self.dst, self.group = accumulators, dst, group
        self.snapshot = torch.zeros_like(accumulators.buffer)
        self.on_gpu = accumulators.buffer.is_cuda
        self.side = torch.cuda.Stream(accumulators.buffer.device) if self.on_gpu else None
        self.work = None

    def submit(self, ordered=False):
        """Snapshot now, reduce in the background.  The snapshot copy runs on torch's current stream: it is ordered after
        the launches already queued only if that is the backend's stream too (``ordered=True``, see
        FusedAccumulators.reduce_to); otherwise the host waits for the integrator first."""
        self.wait()
        if self.acc._backend is not None and not ordered:
            self.acc._backend.synchronize()
        self.snapshot.copy_(self.acc.buffer, non_blocking=True)
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return
        if self.on_gpu:
            self.side.wait_stream(torch.cuda.current_stream(self.snapshot.device))
            with torch.cuda.stream(self.side):
                self.work = dist.reduce(self.snapshot, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            self.work = dist.reduce(self.snapshot, dst=self.dst, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None
        if self.on_gpu:
            self.side.synchronize()

    def images(self):
        """(imageColor [H,W,4], imageRayNb [H,W]) of the last submitted snapshot; the sum over ranks on `dst`."""
        self.wait()
        a = self.acc
        c = self.snapshot[: 4 * a.npix].view(a.height, a.width, 4).cpu().numpy()
        n = self.snapshot[4 * a.npix:].view(a.height, a.width).cpu().numpy()
        return c, n
