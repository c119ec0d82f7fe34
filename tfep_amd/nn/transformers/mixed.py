"""Mixed transformer (reference ``tfep/nn/transformers/mixed.py:29-186``)."""
from typing import Sequence

import torch

from ... import ops
from ...utils.misc import ensure_tensor_sequence
from .transformer import MAFTransformer


class MixedTransformer(MAFTransformer):
    """Apply different transformers to different groups of features.

    ``parameters`` are grouped BY TRANSFORMER (first all parameters of the first transformer,
    ...), each group in that transformer's own layout (reference mixed.py:64-68, :175).
    """

    def __init__(self, transformers: Sequence[MAFTransformer], indices: Sequence[Sequence[int]]):
        super().__init__()
        if len(transformers) < 2:
            raise ValueError('The number of transformers must be greater than 1.')
        if len(transformers) != len(indices):
            raise ValueError('The number of elements in indices must equal that in transformers.')
        # Stored as given, like the reference (mixed.py:58): a plain list keeps the reference's
        # state_dict schema (sub-transformer buffers are then not registered; the spline
        # transformer moves its domain arrays to the input's device on demand).
        self._transformers = transformers
        for idx, ind in enumerate(indices):
            self.register_buffer(f'_indices{idx}', ensure_tensor_sequence(ind))
        par_lengths = [len(t.get_identity_parameters(len(ind))) for t, ind in zip(transformers, indices)]
        self.register_buffer('_parameters_split_indices', torch.cumsum(torch.tensor(par_lengths[:-1]), dim=0))
        self._i32 = {}

    def _apply(self, fn, *args, **kwargs):
        self._i32 = {}
        self.__dict__.pop('_host_splits', None)
        return super()._apply(fn, *args, **kwargs)

    def host_splits(self):
        """Cumulative parameter offsets as a Python list (read from the buffer once: no per-call sync)."""
        h = self.__dict__.get('_host_splits')
        if h is None:
            h = [0] + self._parameters_split_indices.tolist()
            self.__dict__['_host_splits'] = h
        return h

    @property
    def _indices(self):
        return [getattr(self, f'_indices{idx}') for idx in range(len(self._transformers))]

    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        return torch.cat([t.get_identity_parameters(len(ind)).to(torch.get_default_dtype())
                          for t, ind in zip(self._transformers, self._indices)], dim=-1)

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return torch.cat([t.get_degrees_out(degrees_in[ind])
                          for t, ind in zip(self._transformers, self._indices)], dim=-1)

    def forward(self, x, parameters):
        return self._run(x, parameters, inverse=False)

    def inverse(self, y, parameters):
        return self._run(y, parameters, inverse=True)

    def _run(self, x, parameters, inverse):
        ops.check_device_tensor(x, 'x')
        key = str(x.device)
        if key not in self._i32:
            self._i32[key] = [ind.to(device=x.device, dtype=torch.int32) for ind in self._indices]
        y = torch.empty(x.shape, dtype=x.dtype, device=x.device)
        ldj = None
        splits = self.host_splits() + [parameters.shape[1]]
        for t, ind, a, b in zip(self._transformers, self._i32[key], splits[:-1], splits[1:]):
            xg = ops.gather_columns(x, ind)
            par = parameters[:, a:b]
            yg, l = (t.inverse if inverse else t.forward)(xg, par)
            ops.scatter_columns(yg, ind, y)
            ldj = l if ldj is None else ldj + l
        return y, ldj
