"""Oracle (numpy): TFEP KL loss and the FEP estimator.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
Reference files restated here: ``tfep/loss.py`` and ``tfep/analysis/estimator.py``.
"""
import numpy as np


def _logsumexp(a, axis=-1):
    m = np.max(a, axis=axis, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    return (np.log(np.sum(np.exp(a - m), axis=axis, keepdims=True)) + m).squeeze(axis)


def _softmax(a):
    e = np.exp(a - np.max(a))
    return e / np.sum(e)


def boltzmann_kl_div_loss(target_potentials, log_det_J=None, log_weights=None,
                          ref_potentials=None, ignore_nan=False):
    """BoltzmannKLDivLoss.forward.  Ref: loss.py:125-140.

    ``r = u_B - ldj [- u_A]``; unweighted ``mean(r)`` (``nanmean``); weighted
    ``sum(softmax(log_w) * r)`` (``nansum``) with the softmax over the whole batch.
    """
    r = np.asarray(target_potentials)
    if log_det_J is not None:
        r = r - log_det_J
    if ref_potentials is not None:
        r = r - ref_potentials
    if log_weights is not None:
        w = _softmax(np.asarray(log_weights))
        return np.nansum(w * r) if ignore_nan else np.sum(w * r)
    return np.nanmean(r) if ignore_nan else np.mean(r)


def fep_estimator(data, kT=1.0, weights=None, vectorized=False):
    """``dF = -kT logsumexp(-w/kT + log_weights)``.  Ref: analysis/estimator.py:61-86."""
    data = np.asarray(data)
    if vectorized:                                            # estimator.py:62-66
        if data.ndim == 2:
            work, bias = data, None
        else:
            work, bias = np.transpose(data, (2, 0, 1))   # code, not docstring: (n_boot, N, 2)
    else:                                                     # estimator.py:67-71
        if data.ndim == 1:
            work, bias = data, None
        else:
            work, bias = data.T                          # code, not docstring: (N, 2)
    if bias is None:                                          # estimator.py:74-79
        if weights is None:
            log_weights = -np.log(np.asarray(work.shape[-1], dtype=work.dtype))
        else:
            log_weights = np.log(weights)
    elif weights is not None:
        raise NotImplementedError('Bayesian bootstrapping is not supported with biased data.')
    else:                                                     # estimator.py:84 log_softmax
        b = bias / kT
        log_weights = b - _logsumexp(b, axis=-1)[..., None]
    return -kT * _logsumexp(-work / kT + log_weights, axis=-1)
