"""Multi-GPU plumbing: one process per GPU, ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI).

The flow is row-wise independent, so the batch shards across ranks with NO collective on the
data path.  The only cross-sample coupling is in the TFEP reductions; their sufficient statistics
(9 float64, see tfep_hip.h: counts, sums and (max, rescaled-sum) pairs) combine with a single
all-gather of 9 scalars per rank followed by a local (max, rescale, sum) combine -- latency only.
"""
import torch
import torch.distributed as dist

# layout of the statistics vector (tfep_hip.h)
_PLAIN = (0, 1)                       # count, sum r
_PAIRS = ((2, (3, 4)), (5, (6,)), (7, (8,)))   # (max index, indices of sums rescaled by exp(max - global max))


def combine_stats(stats_list):
    """Combine per-shard statistics ``(n_shards, 9)`` into the statistics of the union."""
    s = torch.stack(list(stats_list)) if not torch.is_tensor(stats_list) else stats_list
    out = torch.empty(9, dtype=s.dtype, device=s.device)
    for i in _PLAIN:
        out[i] = s[:, i].sum()
    for mi, sums in _PAIRS:
        m = s[:, mi]
        gm = torch.where(torch.isnan(m).any(), torch.full_like(m[0], float('nan')), m.max())
        scale = torch.exp(m - gm)
        scale = torch.where(torch.isinf(m) & (m < 0), torch.zeros_like(scale), scale)   # empty shard
        out[mi] = gm
        for si in sums:
            out[si] = (s[:, si] * scale).sum()
    return out


def allreduce_stats(stats, group=None):
    """All ranks get the statistics of the global batch (no-op without an initialised group)."""
    if not (dist.is_available() and dist.is_initialized()):
        return stats
    world = dist.get_world_size(group)
    gathered = [torch.empty_like(stats) for _ in range(world)]
    dist.all_gather(gathered, stats, group=group)
    return combine_stats(gathered)


def shard_rows(n_rows, rank, world_size):
    """Contiguous row block ``[begin, end)`` of this rank (remainder spread over the first ranks)."""
    base, rem = divmod(n_rows, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


#: Gradients below this size are packed into one small flat buffer per call (a handful of biases and weight-norm gains: a
#: collective per 60 KB tensor would be latency only); everything larger is reduced IN PLACE.
SMALL_GRADIENT_BYTES = 1 << 20


def _reduce_small(grads, group, world, average):
    """One all-reduce for the small gradients of a call (copied into a flat buffer and back: a few MB at cfg2)."""
    if not grads:
        return
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    if average:
        flat /= world
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n


def allreduce_gradients(module, group=None, bucket_bytes=None, average=False, reduce_scatter=None, small_bytes=None):
    """Data-parallel training: SUM the ``.grad`` of every parameter across ranks, between ``loss.backward()`` and
    ``optimizer.step()``.

    The sum is what pairs with ``BoltzmannKLDivLoss(distributed=True)``: that loss is the loss of the GLOBAL batch and
    its backward already weights every local sample by ``1/N_global`` (or by the global softmax weight), so the
    gradient of the reported loss is the sum of the ranks' local gradients.  ``average=True`` divides by the world
    size afterwards -- only for a loss that each rank normalises by its LOCAL batch (``distributed=False``).

    NO staging copy (round 4; rounds 1-3 concatenated 1 GiB buckets: a 22 GB copy out and back per cfg2 step): every gradient
    of ``small_bytes`` (default 1 MiB) or more -- at cfg2 the twelve weight_v gradients of 0.18 / 0.9 / 4.5 GB, each long
    enough to run at link bandwidth on the point-to-point xGMI fabric -- is all-reduced IN PLACE, all of them queued
    asynchronously before the first wait; the small ones (biases, weight-norm gains) share one flat buffer.  RCCL's
    all-reduce is itself the reduce-scatter + all-gather ring.  ``bucket_bytes`` / ``reduce_scatter`` are accepted for
    compatibility and ignored.  To overlap the collectives with the backward itself use ``OverlappedGradientSync``.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    small_bytes = SMALL_GRADIENT_BYTES if small_bytes is None else small_bytes
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    big = [g for g in grads if g.numel() * g.element_size() >= small_bytes and g.is_contiguous()]
    big_ids = {id(g) for g in big}
    handles = [dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group, async_op=True) for g in big]
    _reduce_small([g for g in grads if id(g) not in big_ids], group, world, average)
    for h in handles:
        h.wait()
    if average:
        for g in big:
            g /= world


class OverlappedGradientSync:
    """Gradient all-reduce that OVERLAPS the backward: a post-accumulate hook on every parameter queues the (asynchronous,
    in-place) all-reduce of a large gradient the moment autograd has produced it, so the collective of MAF layer k travels
    over xGMI while layer k - 1's backward kernels run (one autograd node per MAF layer, ``flows/_backward.py``: a layer's
    gradients appear together).  ``wait()`` -- between ``loss.backward()`` and ``optimizer.step()`` -- reduces the small
    gradients in one flat buffer and waits for everything in flight.  Sums, like ``allreduce_gradients``
    (``average=True`` divides by the world size).  No-op without an initialised process group.

        sync = OverlappedGradientSync(flow)
        for batch in loader:
            optimizer.zero_grad(set_to_none=True)
            loss_fn(*flow(batch)).backward()      # collectives start inside
            sync.wait()
            optimizer.step()
    """

    def __init__(self, module, group=None, average=False, small_bytes=None):
        self.group, self.average = group, average
        self.small_bytes = SMALL_GRADIENT_BYTES if small_bytes is None else small_bytes
        self._pending, self._small = [], []
        self.launched_in_backward = 0            # (diagnostics / tests: collectives queued by the hooks since the last wait)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for p in module.parameters() if p.requires_grad]

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def _on_grad(self, p):
        if not self._active() or p.grad is None:
            return
        g = p.grad
        if g.numel() * g.element_size() >= self.small_bytes and g.is_contiguous():
            self._pending.append((g, dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
            self.launched_in_backward += 1
        else:
            self._small.append(g)

    def wait(self):
        if not self._active():
            self._pending, self._small = [], []
            return
        world = dist.get_world_size(self.group)
        _reduce_small(self._small, self.group, world, self.average)
        for g, h in self._pending:
            h.wait()
            if self.average:
                g /= world
        self._pending, self._small = [], []
        self.launched_in_backward = 0

    def remove(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
