"""(T)FEP free-energy estimator (reference ``tfep/analysis/estimator.py:24-86``)."""
import torch

from ..distributed import allreduce_stats
from ..loss import reduce_stats


def fep_estimator(data, kT=1.0, weights=None, vectorized=False, process_group=None, distributed=False):
    r"""``dF = -kT logsumexp(-w/kT + log_weights)``.

    ``data`` is ``(n_samples,)`` work values, or ``(n_samples, 2)`` with the bias (log-weight)
    in the second column -- the layout the reference CODE accepts (estimator.py:67-71; its
    docstring says ``(2, n_samples)``).  ``vectorized=True`` adds a leading bootstrap dimension:
    ``(n_bootstraps, n_samples)`` or ``(n_bootstraps, n_samples, 2)``; ``weights``
    ``(n_bootstraps, n_samples)`` are Bayesian-bootstrap weights.

    With ``distributed=True`` (or a ``process_group``) ``data`` is this rank's shard and the
    estimate is of the global sample: one all-reduce of the (max, sum-exp) statistics.
    """
    if not vectorized:
        if weights is not None:
            weights = weights[None]
        return _estimate(data[None], kT, weights, process_group, distributed)[0]
    return _estimate(data, kT, weights, process_group, distributed)


def _estimate(data, kT, weights, process_group, distributed):
    distributed = distributed or process_group is not None
    if data.dim() == 2:
        work, bias = data, None
    else:
        work, bias = data[..., 0], data[..., 1]
    if bias is not None and weights is not None:
        raise NotImplementedError('Bayesian bootstrapping is not supported with biased data.')
    out = []
    for b in range(work.shape[0]):
        w = work[b].contiguous().float()
        # the bias slot of tfep_tfep_reduce is  + bias/kT ; weights enter as log(weights)*kT
        if bias is not None:
            extra = bias[b].contiguous().float()
        elif weights is not None:
            extra = torch.log(weights[b].contiguous().float()) * kT
        else:
            extra = None
        stats = reduce_stats(w, None, None, None, extra, kT=kT)       # torch.ops.tfep.tfep_reduce
        if distributed:
            stats = allreduce_stats(stats, process_group)
        lse = stats[5] + torch.log(stats[6])                    # logsumexp(-w/kT [+ bias/kT])
        if bias is not None:
            lse = lse - (stats[7] + torch.log(stats[8]))        # - logsumexp(bias/kT)  (log_softmax)
        elif weights is None:
            lse = lse - torch.log(stats[0])                     # - log N
        out.append(-kT * lse)
    return torch.stack(out).to(data.dtype)
