"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The hot path shards naturally: rays and images are independent, so the forward
needs no collective.  Training adds exactly one exchange per step: a sum all-reduce of the
parameter gradients (2 x 1 518 979 fp32 = 12.15 MB for the two MLPs).  xGMI is point-to-point
(7 links per GPU), so the gradients go out as ONE flat bucket per stream rather than 48 small
tensors: a ring all-reduce is per-link bound and small messages only pay latency.

``shard_rays`` / ``shard_images`` give each rank its slice; ``GradAllReducer`` does the exchange.
Everything here also runs on the ``gloo`` backend (CPU tensors), which is how it is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Sequence

import torch


def shard_range(n: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of n items for `rank`; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_images(batch: int, rank: int, world: int):
    """Data-parallel by image (cfg4: global batch 16 -> 2 images per rank on 8 ranks)."""
    return shard_range(batch, rank, world)


def shard_rays(batch_xy: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """B=1 inference: contiguous row blocks of the ray grid, [B,2,N_r] -> [B,2,N_r/world]."""
    lo, hi = shard_range(batch_xy.shape[-1], rank, world)
    return batch_xy[:, :, lo:hi].contiguous()


def gather_rays(local: torch.Tensor, n_rays: int, world: int) -> torch.Tensor:
    """Inverse of shard_rays for an output [B,C,N_r_local]: all_gather along the ray axis (only
    needed when one rank wants the full map; the op itself never calls it)."""
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_range(n_rays, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros(*local.shape[:-1], mx, dtype=local.dtype, device=local.device)
    pad[..., :local.shape[-1]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[..., :hi - lo] for b, (lo, hi) in zip(bufs, sizes)], dim=-1)


class GradAllReducer:
    """Sum-all-reduce (then average) of the .grad of `params` in flat buckets.

    ``params`` may be a flat sequence (cut into buckets of >= bucket_numel elements: by default one MLP stream,
    1 518 979 floats = 6 MB) or a list of lists (explicit buckets).  xGMI is point-to-point, a ring all-reduce is
    per-link bound and small messages only pay latency: few flat buckets, not one message per tensor.

    Overlap: ``arm_overlap()`` registers post-accumulate-grad hooks; a bucket whose gradients are all written is
    flattened and its all-reduce launched at once on the communication stream, while the autograd engine is still
    working on the layers before it (in the whole network: the NeuralRenderer bucket flies during the hot path's
    backward).  The two MLP buckets come out of ONE gnr_bwd call together, so they start back to back at its end:
    there is nothing left to overlap them with.  ``all_reduce()`` launches whatever has not been launched, waits,
    and writes the averaged gradients back.  ``bytes_per_step`` / ``n_buckets`` describe the exchange.

    Assumptions (asserted where they can be): every rank owns the same parameters in the same buckets and EVERY parameter
    of a bucket receives a gradient in every backward -- a rank whose data-dependent control flow skips a parameter
    would launch a different sequence of collectives than its peers and hang them (the reference network has no such
    parameter: every tensor of the two MLPs and of the NeuralRenderer is used by every step;
    ``all_reduce(check_complete=True)`` verifies that no bucket was left half counted).  Gradient accumulation over several
    backward passes: wrap all but the last one in ``no_sync()`` (hooks then launch nothing), or call ``begin_step()``
    where the loop zeroes the gradients.  Without either, a bucket whose exchange is already in flight when its hooks
    fire again is collected and re-launched -- correct, but it blocks the autograd thread on a stale collective.

    ``force_collective``: run the exchange even at world size 1 (``bench.py`` with GNR_BENCH_FORCE_DIST=1: a one-rank RCCL
    communicator exercises library load, communicator creation and the stream hand-off on a single GPU)."""

    def __init__(self, params, world_size: int, bucket_numel: int = 1518979, average: bool = True,
                 force_collective: bool = False):
        self.world = world_size
        self.average = average
        self.active = world_size > 1 or force_collective
        self._sync = True
        self.buckets: List[List[torch.Tensor]] = []
        params = list(params)
        if params and isinstance(params[0], (list, tuple)):
            self.buckets = [list(b) for b in params if len(b)]
        else:
            cur, n = [], 0
            for p in params:
                cur.append(p)
                n += p.numel()
                if n >= bucket_numel:
                    self.buckets.append(cur)
                    cur, n = [], 0
            if cur:
                self.buckets.append(cur)
        self.params: List[torch.Tensor] = [p for b in self.buckets for p in b]
        self.n_buckets = len(self.buckets)
        self.bytes_per_step = sum(p.numel() for p in self.params) * 4
        self._inflight = {}
        self._seen = None            # per bucket: ids of the parameters whose hook fired since the last (re)start
        self._hooks = []
        self.relaunched = 0          # buckets whose exchange was launched twice (accumulation): a statistic

    # -- overlap -------------------------------------------------------------------------------------------
    def arm_overlap(self):
        """Launch a bucket's all-reduce from autograd hooks as soon as its last gradient has been accumulated."""
        if not self.active or self._hooks:
            return
        self._seen = [set() for _ in self.buckets]
        for bi, bucket in enumerate(self.buckets):
            for p in bucket:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def begin_step(self):
        """Forget partly counted buckets and drop exchanges that were launched but never collected (call it where the
        training loop zeroes the gradients, or after a step that raised)."""
        for bi in list(self._inflight):
            work, _ = self._inflight.pop(bi)
            work.wait()
        if self._seen is not None:
            self._seen = [set() for _ in self.buckets]

    def no_sync(self):
        """Context manager for the accumulation micro-steps of a step (all backward passes but the last): the hooks count
        nothing and launch nothing, so no collective is started on gradients that are still going to change."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield self
            finally:
                self._sync = old
        return ctx()

    def _make_hook(self, bi):
        def hook(param):
            if not self._sync:
                return
            seen = self._seen[bi]
            if bi in self._inflight:
                # a second backward before all_reduce(): the flat copy in flight is stale.  Collect it (every rank
                # does, in the same order) and count this backward's gradients afresh.
                work, _ = self._inflight.pop(bi)
                work.wait()
                seen.clear()
                self.relaunched += 1
            elif id(param) in seen:
                seen.clear()         # the previous backward never finished this bucket
            seen.add(id(param))
            if len(seen) == len(self.buckets[bi]):
                seen.clear()
                self._launch(bi)
        return hook

    def _launch(self, bi):
        import torch.distributed as dist
        bucket = self.buckets[bi]
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
        flat = torch.cat([g.reshape(-1) for g in grads])
        self._inflight[bi] = (dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat)

    # -- the exchange --------------------------------------------------------------------------------------
    def all_reduce(self, check_complete: bool = False):
        if not self.active:
            return
        if check_complete and self._seen is not None:
            half = [bi for bi, seen in enumerate(self._seen) if seen and bi not in self._inflight]
            if half:
                raise RuntimeError("GradAllReducer: bucket(s) %s received gradients for only part of their parameters in "
                                   "this backward: a parameter without a gradient breaks the collective order" % half)
        for bi in range(self.n_buckets):
            if bi not in self._inflight:
                self._launch(bi)
        for bi in range(self.n_buckets):
            work, flat = self._inflight.pop(bi)
            work.wait()
            if self.average:
                flat.div_(self.world)
            off = 0
            for p in self.buckets[bi]:
                n = p.numel()
                g = flat[off:off + n].view_as(p)
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    p.grad.copy_(g)
                off += n
        if self._seen is not None:
            self._seen = [set() for _ in self.buckets]
