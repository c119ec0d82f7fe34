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


def allreduce_gradients(module, group=None, bucket_bytes=1 << 30, average=False, reduce_scatter=None):
    """Data-parallel training: SUM the ``.grad`` of every parameter across ranks.

    The sum is what pairs with ``BoltzmannKLDivLoss(distributed=True)``: that loss is the loss of the GLOBAL batch and
    its backward already weights every local sample by ``1/N_global`` (or by the global softmax weight), so the
    gradient of the reported loss is the sum of the ranks' local gradients.  ``average=True`` divides by the world
    size afterwards -- only for a loss that each rank normalises by its LOCAL batch (``distributed=False``).

    Gradients are copied into a few large flat buckets (default 1 GiB: the 22 GB of cfg2 gradients
    go out as ~22 collectives, each long enough to run at link bandwidth on the point-to-point
    xGMI fabric).  On RCCL (``backend='nccl'``) each bucket goes out as an explicit reduce-scatter followed by an
    all-gather (``reduce_scatter=None``: chosen when the backend supports it) -- the two halves of a ring all-reduce,
    with the averaging applied to the 1/world shard between them; ``gloo`` (CPU tests) uses ``all_reduce``.
    Call between ``loss.backward()`` and ``optimizer.step()``.
    """
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if reduce_scatter is None:
        reduce_scatter = dist.get_backend(group) == 'nccl'
    bucket, size = [], 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([g.reshape(-1) for g in bucket])
        if reduce_scatter:
            n = flat.numel()
            padded = (n + world - 1) // world * world
            if padded != n:
                flat = torch.cat([flat, flat.new_zeros(padded - n)])
            shard = torch.empty(padded // world, dtype=flat.dtype, device=flat.device)
            dist.reduce_scatter_tensor(shard, flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                shard /= world
            dist.all_gather_into_tensor(flat, shard, group=group)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                flat /= world
        off = 0
        for g in bucket:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n
        bucket, size = [], 0

    for g in grads:
        nbytes = g.numel() * g.element_size()
        if size + nbytes > bucket_bytes and bucket:
            flush()
        bucket.append(g)
        size += nbytes
    flush()
