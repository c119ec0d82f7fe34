"""Gaussian radial bases with the ``tfep.nn.embeddings.radial`` API (reference ``tfep/nn/embeddings/radial.py``).

``GaussianBasisExpansion`` (:24-137), ``behler_parrinello_cosine_switching_function`` (:144-176) and
``BehlerParrinelloRadialExpansion`` (:179-291): same constructors, ``from_range`` and ``state_dict`` (only
``_log_gammas`` / ``_means`` when trainable; the fixed ones are non-persistent buffers here so that ``.to(device)``
moves them -- the reference keeps plain tensor attributes).  ``forward`` is ``tfep_radial_expansion``; inside
``EGNNDynamics`` the expansion is fused into the edge kernel and never materialised.
"""
import math

import torch

from ... import _lib


class GaussianBasisExpansion(torch.nn.Module):
    """Soft one-hot encoding of a scalar on a Gaussian basis: ``exp(-gamma_k (x - mean_k)^2)``."""

    def __init__(self, means, stds, trainable_means=False, trainable_stds=False):
        super().__init__()
        log_gammas = torch.log(1 / stds ** 2)            # inverse variances in log units: always positive when trained
        if trainable_means:
            self._means = torch.nn.Parameter(means)
        else:
            self.register_buffer('_means', means, persistent=False)
        if trainable_stds:
            self._log_gammas = torch.nn.Parameter(log_gammas)
        else:
            self.register_buffer('_log_gammas', log_gammas, persistent=False)

    @classmethod
    def from_range(cls, n_gaussians, max_mean, min_mean=0.0, relative_std=3.0, trainable_means=False,
                   trainable_stds=False):
        """Equidistant Gaussians on ``[min_mean, max_mean]`` with std = ``relative_std`` x spacing."""
        means, stds = cls._get_equidistant_means_and_stds(n_gaussians, max_mean, min_mean, relative_std)
        return cls(means, stds, trainable_means=trainable_means, trainable_stds=trainable_stds)

    def _expand(self, data, r_cutoff=0.0, switching=False, force_zero=True):
        _lib.check_device_tensor(data, 'data')
        if data.shape[-1] == 1:
            data = data.squeeze(-1)
        flat = data.contiguous().reshape(-1)
        means = self._means.detach().to(device=flat.device, dtype=torch.float32).contiguous()
        lg = self._log_gammas.detach().to(device=flat.device, dtype=torch.float32).contiguous()
        out = torch.empty(flat.numel(), len(means), dtype=torch.float32, device=flat.device)
        _lib.call('tfep_radial_expansion', _lib.ptr(flat), flat.numel(), _lib.ptr(means), _lib.ptr(lg), len(means),
                  float(r_cutoff), int(switching), int(force_zero), _lib.ptr(out), _lib.stream_of(flat))
        return out.reshape(*data.shape, len(means))

    def forward(self, data):
        """``(batch, *) -> (batch, *, n_gaussians)``."""
        return self._expand(data)

    @classmethod
    def _get_equidistant_means_and_stds(cls, n_gaussians, max_mean, min_mean, relative_std):
        spacing = (max_mean - min_mean) / (n_gaussians - 1)
        means = torch.linspace(min_mean, max_mean, n_gaussians)
        stds = torch.full((len(means),), fill_value=relative_std * spacing)
        return means, stds


def behler_parrinello_cosine_switching_function(r_cutoff, r, force_zero_after_cutoff=True):
    """``0.5 cos(pi r / r_cutoff) + 0.5``, zero beyond the cutoff (elementwise; any device)."""
    value = 0.5 * torch.cos(math.pi / r_cutoff * r) + 0.5
    if force_zero_after_cutoff:
        value = torch.where(r > r_cutoff, torch.zeros_like(value), value)
    return value


class BehlerParrinelloRadialExpansion(GaussianBasisExpansion):
    """Gaussian basis times the Behler-Parrinello cosine switching function."""

    def __init__(self, r_cutoff, means, stds, trainable_means=False, trainable_stds=False,
                 force_zero_after_cutoff=True):
        super().__init__(means, stds, trainable_means, trainable_stds)
        self.r_cutoff = r_cutoff
        self.force_zero_after_cutoff = force_zero_after_cutoff

    @classmethod
    def from_range(cls, r_cutoff, n_gaussians, max_mean, min_mean=0.0, relative_std=3.0, trainable_means=False,
                   trainable_stds=False, force_zero_after_cutoff=True):
        means, stds = cls._get_equidistant_means_and_stds(n_gaussians, max_mean, min_mean, relative_std)
        # (like the reference, radial.py:263-266: force_zero_after_cutoff is not forwarded, the default True applies)
        return cls(r_cutoff, means, stds, trainable_means=trainable_means, trainable_stds=trainable_stds)

    def forward(self, distances):
        return self._expand(distances, self.r_cutoff, switching=True, force_zero=self.force_zero_after_cutoff)
