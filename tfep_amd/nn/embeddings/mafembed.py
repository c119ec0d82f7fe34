"""MAF embedding layers (reference ``tfep/nn/embeddings/mafembed.py``): periodic embedding."""
import abc
from typing import Optional, Sequence

import torch

from ... import ops
from ...utils.misc import ensure_tensor_sequence, remove_and_shift_sorted_indices


class MAFEmbedding(abc.ABC, torch.nn.Module):
    """Embedding applied to the conditioner input of a MAF (reference mafembed.py:31-60)."""

    @abc.abstractmethod
    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        pass


class PeriodicEmbedding(MAFEmbedding):
    """Lift periodic features to ``(cos, sin)`` (reference mafembed.py:65-167).

    Output layout follows the reference CODE: non-periodic features first, then the
    ``cos, sin`` pairs of the periodic ones (mafembed.py:137-145).
    """

    def __init__(self, n_features_in: int, limits: Sequence[float],
                 periodic_indices: Optional[Sequence[int]] = None):
        super().__init__()
        self.register_buffer('limits', ensure_tensor_sequence(limits))
        if periodic_indices is None:
            periodic_indices = torch.arange(n_features_in)
        else:
            periodic_indices = ensure_tensor_sequence(periodic_indices)
            if len(periodic_indices.unique()) < len(periodic_indices):
                raise ValueError('Found duplicated indices in periodic_indices.')
        self.register_buffer('_periodic_indices', periodic_indices)
        self.register_buffer('_nonperiodic_indices', remove_and_shift_sorted_indices(
            indices=torch.arange(n_features_in), removed_indices=periodic_indices, shift=False))
        self._i32 = {}

    def _apply(self, fn, *args, **kwargs):
        self._i32 = {}
        self.__dict__.pop('_host_limits', None)
        return super()._apply(fn, *args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        self.__dict__.pop('_host_limits', None)
        return super()._load_from_state_dict(*args, **kwargs)

    def host_limits(self):
        """(lower, upper) as Python floats, read from the ``limits`` buffer once (no per-call sync)."""
        h = self.__dict__.get('_host_limits')
        if h is None:
            h = (float(self.limits[0]), float(self.limits[1]))
            self.__dict__['_host_limits'] = h
        return h

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops.check_device_tensor(x, 'x')
        key = str(x.device)
        if key not in self._i32:
            self._i32[key] = (self._periodic_indices.to(device=x.device, dtype=torch.int32),
                              self._nonperiodic_indices.to(device=x.device, dtype=torch.int32))
        per, non = self._i32[key]
        return ops.periodic_embedding(x, per, non, *self.host_limits())

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return torch.cat([degrees_in[self._nonperiodic_indices],
                          degrees_in[self._periodic_indices].repeat_interleave(2)])
