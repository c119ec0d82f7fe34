"""Bootstrap analysis of a statistic (reference ``tfep/analysis/bootstrap.py:24-262``).

Same API and results as the reference.  When the statistic is :func:`tfep_amd.analysis.fep_estimator` the whole
bootstrap distribution is computed by one HIP kernel (``tfep_bootstrap_fep``: one workgroup per resample, the
resampled data are never materialised); any other statistic follows the reference's gather-and-call scheme with
torch ops on the data's device.
"""
import functools

import torch

from .. import _lib
from .estimator import fep_estimator


def bootstrap(data, statistic, *, confidence_level=0.95, n_resamples=9999, bootstrap_sample_size=None,
              take_first_only=False, batch=None, method='percentile', bayesian=False, generator=None):
    """Parameters of the bootstrap distribution of ``statistic`` over ``data`` ``(n_samples,)`` or
    ``(n_samples, data_dimension)``: confidence interval, standard deviation, mean and median; a list of such
    dicts if ``bootstrap_sample_size`` is a list.  Arguments as reference bootstrap.py:24-121.

    ``generator`` may live on the CPU (the resample indices are then drawn exactly as the reference draws them and
    moved to the data's device) or on the device.
    """
    n_samples = len(data)
    if bayesian and generator is not None:
        raise ValueError('Bayesian bootstrapping does not support random number generators.')
    if bootstrap_sample_size is None:
        sizes = [n_samples]
    else:
        if bayesian and not take_first_only:
            raise ValueError('With Bayesian bootstrapping, specifying a bootstrap_sample_size '
                             'is supported only when take_first_only is True.')
        sizes = [bootstrap_sample_size] if isinstance(bootstrap_sample_size, int) else list(bootstrap_sample_size)
    single = len(sizes) == 1                  # also a one-element list (reference bootstrap.py: len(...) == 1)
    if method not in ('percentile', 'basic'):
        raise ValueError("method must be 'percentile' or 'basic'")
    if isinstance(generator, int):
        generator = torch.Generator(device=data.device).manual_seed(generator)
    if batch is None:
        batch = n_resamples
    kT = _fep_statistic_kT(statistic)

    results = []
    with torch.no_grad():
        for sample_size in sizes:
            stats = torch.empty(n_resamples, dtype=torch.float64 if kT is not None else data.dtype, device=data.device)
            for k in range(0, n_resamples, batch):
                nb = min(batch, n_resamples - k)
                if bayesian:
                    weights = torch.distributions.Dirichlet(
                        torch.ones(sample_size, device=data.device)).sample((nb,))
                    sub = data[:sample_size]
                    if kT is not None:
                        stats[k:k + nb] = bootstrap_fep(sub, weights=weights, kT=kT)
                    else:
                        stats[k:k + nb] = statistic(sub.expand((nb, *sub.shape)), weights=weights, vectorized=True)
                else:
                    high = sample_size if take_first_only else n_samples
                    gen_device = generator.device if generator is not None else data.device
                    idx = torch.randint(low=0, high=high, size=(nb, sample_size), generator=generator,
                                        device=gen_device).to(data.device)
                    if kT is not None:
                        stats[k:k + nb] = bootstrap_fep(data, indices=idx, kT=kT)
                    else:
                        samples = data[idx]                       # (nb, sample_size[, data_dimension])
                        stats[k:k + nb] = statistic(samples, vectorized=True)
            stats = stats.to(data.dtype)
            alpha = (1 - confidence_level) / 2
            ci_l, ci_u = torch.quantile(stats, q=torch.tensor([alpha, 1 - alpha], dtype=stats.dtype, device=stats.device))
            if method == 'basic':
                # the reference calls statistic(data.unsqueeze(0)) (bootstrap.py:170), which its own fep_estimator
                # rejects; the estimator gets the vectorized call it needs, other statistics the reference's call
                full = (statistic(data.unsqueeze(0), vectorized=True) if kT is not None
                        else statistic(data.unsqueeze(0))).reshape(())
                ci_l, ci_u = 2 * full - ci_u, 2 * full - ci_l
            results.append(dict(confidence_interval=dict(low=ci_l, high=ci_u), standard_deviation=torch.std(stats),
                                mean=torch.mean(stats), median=torch.median(stats)))
    return results[0] if single else results


def _fep_statistic_kT(statistic):
    """kT if ``statistic`` is fep_estimator (possibly a functools.partial fixing only kT), else None."""
    if statistic is fep_estimator:
        return 1.0
    if isinstance(statistic, functools.partial) and statistic.func is fep_estimator and not statistic.args \
            and set(statistic.keywords) <= {'kT'}:
        return float(statistic.keywords.get('kT', 1.0))
    return None


def bootstrap_fep(data, indices=None, weights=None, kT=1.0):
    """``fep_estimator`` of every resample in one kernel: ``data`` is ``(n_samples,)`` work values or
    ``(n_samples, 2)`` with the bias in the second column; ``indices`` ``(n_resamples, sample_size)`` int64 picks the
    resamples (standard bootstrap), or ``weights`` ``(n_resamples, n_samples)`` are Bayesian-bootstrap weights.
    Returns ``(n_resamples,)`` float64."""
    if (indices is None) == (weights is None):
        raise ValueError('exactly one of indices and weights must be given')
    if data.dim() == 2:
        work, bias = data[:, 0].contiguous().float(), data[:, 1].contiguous().float()
    else:
        work, bias = data.contiguous().float(), None
    _lib.check_device_tensor(work, 'data')
    if indices is not None:
        indices = indices.to(device=work.device, dtype=torch.int64).contiguous()
        R, S = indices.shape
    else:
        if bias is not None:
            raise NotImplementedError('Bayesian bootstrapping is not supported with biased data.')
        weights = weights.to(device=work.device, dtype=torch.float32).contiguous()
        R, S = weights.shape
        if S != work.numel():
            raise ValueError('weights must have one column per sample')
    out = torch.empty(R, dtype=torch.float64, device=work.device)
    _lib.call('tfep_bootstrap_fep', _lib.ptr(work), _lib.ptr(bias), _lib.ptr(indices), _lib.ptr(weights), work.numel(),
              R, S, float(kT), _lib.ptr(out), _lib.stream_of(work))
    return out
