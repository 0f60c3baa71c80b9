"""Source-sharded registration: one process per GPU, target replicated.

Every rank holds the whole target (its BVH and tables are built independently
from identical input, so they are identical) and its shard of the kept source
points: chunks of the kept points' order along a space-filling curve, dealt out
to the ranks in turn (icpx_set_shard).  One iteration is: local fused search +
fp64 partial sums on the GPU, ONE all-reduce of the 40-double vector (35 sums
and 5 job-wide counts the host's policy goes by; RCCL over xGMI when the
process group's backend is "nccl"; the message is 320 bytes, so the collective
is latency-bound and there is exactly one per iteration), then the same solve
on every rank's device, which keeps the transforms identical without a
broadcast.  The reference has no multi-process path; the loop is
vtkICP::InternalUpdate's (ICP/vtkICP.cxx:189-246).
"""
import numpy as np

from . import icpx
from .vtkicp import icp_loop


def shard_bounds(n_kept, rank, world):
    """Block [k0, k1) of `rank` when n_kept items are cut into `world` equal blocks in their
    own order (icpx_set_shard's split with shards by blocks of kept indices, and what a host
    loop over its own arrays uses)."""
    return n_kept * rank // world, n_kept * (rank + 1) // world


def shard_size(n_kept, rank, world):
    """Number of kept points icpx_set_shard(rank, world) gives this rank: chunks of the kept
    points' order along the space-filling curve (8192 points; halved while a rank would have
    fewer than eight) are dealt out to the ranks in turn."""
    if world == 1:
        return n_kept
    chunk = 8192
    while chunk > 1 and n_kept < 8 * world * chunk:
        chunk //= 2
    nchunks = (n_kept + chunk - 1) // chunk
    mine = (nchunks - rank + world - 1) // world if rank < nchunks else 0
    n = mine * chunk
    if mine > 0 and (nchunks - 1) % world == rank and n_kept % chunk:
        n -= chunk - n_kept % chunk
    return n


def all_reduce_device(t, group=None):
    """Sum-all-reduce of a small device tensor, ordered on the current stream.  With RCCL
    (backend "nccl") the collective is a kernel on the device.  Any other backend gets the
    tensor through a host copy made here: gloo's own handling of device tensors (side
    streams plus a host wait) stalls when two ranks share one GPU on this ROCm build, with
    or without this library's kernels in flight (tools/gloo_probe.py shows it with plain
    torch ops), so it is never handed one."""
    import torch
    import torch.distributed as dist

    if dist.get_backend(group) == "nccl":
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return
    host = t.to("cpu")  # stream-ordered copy; returns when the data is on the host
    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
    t.copy_(host)


def make_allreduce(group=None, device_tensor=None):
    """Returns allreduce(sums) for the default/current torch.distributed group, or None when
    the job has a single rank.  With `device_tensor` (a CUDA float64 tensor the sums were
    written into) the reduction runs on the device through RCCL; otherwise on host arrays."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if device_tensor is None:
            return None
        return lambda _s: device_tensor.cpu().numpy()

    if device_tensor is not None:
        def reduce_dev(_sums):
            all_reduce_device(device_tensor, group)
            return device_tensor.cpu().numpy()
        return reduce_dev

    def reduce_host(sums):
        t = torch.from_numpy(np.ascontiguousarray(sums, np.float64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.numpy()
    return reduce_host


class ShardedICP:
    """One rank's share of a registration driven through torch.distributed."""

    def __init__(self, device=0, rank=0, world=1, group=None, use_device_allreduce=True):
        import torch

        self.rank, self.world, self.group = rank, world, group
        self.exit_on_error = True  # multi-rank runs: a local failure ends this process
        self.ctx = icpx.Context(device)
        self.ctx.set_shard(rank, world)
        self.sums_dev = None
        self.stream = None
        if use_device_allreduce:
            # The library and the collective share a stream of their own rather than the
            # default (null) stream and its implicit synchronisation with other streams.
            torch.cuda.set_device(device)
            self.stream = torch.cuda.Stream(device=device)
            with torch.cuda.stream(self.stream):
                self.sums_dev = torch.zeros(icpx.NSUMS, dtype=torch.float64,
                                            device="cuda:%d" % device)
            self.ctx.set_stream(self.stream.cuda_stream)

    def set_target(self, xyz, tri, cell_normals=None):
        self.ctx.set_target(xyz, tri, cell_normals)

    def set_source(self, xyz, normals=None):
        self.ctx.set_source(xyz, normals)

    def iterate(self, metric, T, step, max_dist, min_dot=0.0):
        """One search + partial sums + all-reduce; returns the global sums (numpy)."""
        if self.sums_dev is not None:
            import torch

            self.ctx.accumulate(metric, T, step, max_dist, min_dot,
                                device_ptr=self.sums_dev.data_ptr())
            with torch.cuda.stream(self.stream):
                return make_allreduce(self.group, self.sums_dev)(None)
        sums = self.ctx.accumulate(metric, T, step, max_dist, min_dot)
        ar = make_allreduce(self.group)
        return sums if ar is None else ar(sums)

    def _local_failure(self, exc, multi):
        """A rank whose library call failed must not leave the others waiting in the
        all-reduce: in a multi-rank job this process ends (the peers' collectives then fail
        instead of blocking for ever), and it does so before anything that waits for the
        stream, where a collective that cannot complete may be queued.  A single rank closes
        the run."""
        import os
        import sys
        if multi and self.exit_on_error:
            sys.stderr.write("ShardedICP rank %d: %s; leaving the job\n" % (self.rank, exc))
            sys.stderr.flush()
            os._exit(70)
        try:
            self.ctx.run_end()          # clears the run; waits for what was enqueued
        except Exception:
            pass

    def run_device(self, params, force_collective=False, emulate=False):
        """The whole loop without a host round trip per iteration: search and sums on the
        GPU, all-reduce of the 40 sums on the device (stream-ordered, RCCL), solve and
        transform update on the device.  Needs the device all-reduce buffer (the default).
        Returns (T 4x4, iterations, mean_distance, status), the same on every rank.
        The library keeps the loop's position (icpx_run_advance): a settled pass tried as a
        lean one that some rank could not answer is dropped by all and repeated.
        force_collective runs the all-reduce even in a one-rank group (a test of the stream
        ordering between the library's kernels and the collective); emulate runs one rank's
        share of a multi-rank loop by itself, without a process group (measurements of what
        a rank of an N-rank job does: the registration is then of this shard alone)."""
        import torch
        import torch.distributed as dist

        assert self.sums_dev is not None, "run_device needs use_device_allreduce=True"
        grouped = dist.is_available() and dist.is_initialized()
        multi = grouped and (dist.get_world_size(self.group) > 1 or force_collective)
        if not multi and not emulate:
            # one rank: the library's own loop (the solve rides on the reduction's launch)
            with torch.cuda.stream(self.stream):
                return self.ctx.run(params)
        ptr = self.sums_dev.data_ptr()
        try:
            self.ctx.run_begin(params)
            with torch.cuda.stream(self.stream):
                while self.ctx.run_advance(ptr):
                    if multi:
                        all_reduce_device(self.sums_dev, self.group)
                    self.ctx.run_update(ptr)
        except BaseException as exc:
            self._local_failure(exc, grouped and dist.get_world_size(self.group) > 1)
            raise
        return self.ctx.run_end()

    def run(self, metric=1, max_iterations=50, max_landmarks=200, max_distance=10.0,
            check_mean_distance=False, max_mean_distance=0.01, min_normal_dot=0.0, n_points=None):
        npts = n_points if n_points is not None else self.ctx._np
        step = npts // max_landmarks if npts > max_landmarks else 1
        return icp_loop(
            lambda T: self.iterate(metric, T, step, max_distance, min_normal_dot), None,
            self.ctx.origin, metric, max_iterations, check_mean_distance, max_mean_distance)
