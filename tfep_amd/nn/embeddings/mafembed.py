"""MAF embedding layers (reference ``tfep/nn/embeddings/mafembed.py``): periodic, flip-invariant and mixed embeddings
of the conditioner input.

``PeriodicEmbedding`` runs on its HIP kernels (forward and backward).  ``FlipInvariantEmbedding`` is two 2-layer
perceptrons on 2-8 inputs per vector and ``MixedEmbedding`` is index bookkeeping: both are O(batch x features) work in
front of the MADE GEMMs and are written with ordinary (differentiable) torch ops on the device.
"""
import abc
from typing import Optional, Sequence

import torch

from ... import ops
from ...utils.misc import ensure_tensor_sequence, remove_and_shift_sorted_indices


def _selected_and_rest(n_features_in, selected, what):
    """``(selected, rest)`` index tensors: ``selected`` defaults to every feature and must not repeat an index;
    ``rest`` are the features left alone, in order."""
    if selected is None:
        selected = torch.arange(n_features_in)
    else:
        selected = ensure_tensor_sequence(selected)
        if len(selected.unique()) < len(selected):
            raise ValueError(f'Found duplicated indices in {what}.')
    rest = remove_and_shift_sorted_indices(indices=torch.arange(n_features_in), removed_indices=selected, shift=False)
    return selected, rest


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
        periodic, rest = _selected_and_rest(n_features_in, periodic_indices, 'periodic_indices')
        self.register_buffer('limits', ensure_tensor_sequence(limits))
        self.register_buffer('_periodic_indices', periodic)
        self.register_buffer('_nonperiodic_indices', rest)
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

    def device_indices(self, device):
        key = str(device)
        if key not in self._i32:
            self._i32[key] = (self._periodic_indices.to(device=device, dtype=torch.int32),
                              self._nonperiodic_indices.to(device=device, dtype=torch.int32))
        return self._i32[key]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ops.check_device_tensor(x, 'x')
        per, non = self.device_indices(x.device)
        if torch.is_grad_enabled() and x.requires_grad:
            return _PeriodicEmbeddingFn.apply(x, per, non, *self.host_limits())
        return ops.periodic_embedding(x, per, non, *self.host_limits())

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return torch.cat([degrees_in[self._nonperiodic_indices],
                          degrees_in[self._periodic_indices].repeat_interleave(2)])


class _PeriodicEmbeddingFn(torch.autograd.Function):
    """Differentiable periodic embedding: HIP forward, HIP backward (``tfep_periodic_embedding_backward``)."""

    @staticmethod
    def forward(ctx, x, per, non, lower, upper):
        ctx.save_for_backward(x, per, non)
        ctx.limits = (lower, upper)
        return ops.periodic_embedding(x.detach(), per, non, lower, upper)

    @staticmethod
    def backward(ctx, g):
        from ... import _lib
        x, per, non = ctx.saved_tensors
        x, ldx = _lib.rows(x.detach(), 'x')
        g = g.contiguous()
        B, D = x.shape
        gx = torch.zeros(B, D, dtype=torch.float32, device=x.device)
        if B > 0:
            _lib.call('tfep_periodic_embedding_backward', _lib.ptr(x), ldx, _lib.ptr(per), per.numel(), _lib.ptr(non),
                      non.numel(), *ctx.limits, _lib.ptr(g), g.shape[1], _lib.ptr(gx), D, B, _lib.stream_of(x))
        return gx, None, None, None, None


class FlipInvariantEmbedding(MAFEmbedding):
    """Embed vectors (quaternions by default) into a representation invariant to their sign
    (reference mafembed.py:174-348; Koehler et al. 2023, SI eq. 46):
    ``softmax-weighted sum of net(v) and net(-v)`` with the weights from a second small network.

    Output layout follows the reference: the non-embedded features first, then ``embedding_dimension`` features
    per vector.  Arguments as reference mafembed.py:187-215.
    """

    def __init__(self, n_features_in: int, embedding_dimension: int,
                 embedded_indices: Optional[Sequence[int]] = None, vector_dimension: int = 4,
                 hidden_layer_width: int = 32):
        super().__init__()
        embedded, rest = _selected_and_rest(n_features_in, embedded_indices, 'embedded_indices')

        def perceptron(n_out):          # vector_dimension -> hidden_layer_width -> n_out
            return torch.nn.Sequential(torch.nn.Linear(vector_dimension, hidden_layer_width), torch.nn.ELU(),
                                       torch.nn.Linear(hidden_layer_width, n_out))
        self.embedding_layer = perceptron(embedding_dimension)      # the candidate embeddings of v and -v
        self.weight_layer = perceptron(1)                           # their (pre-softmax) weights
        self.register_buffer('_embedded_indices', embedded)
        self.register_buffer('_nonembedded_indices', rest)

    @property
    def vector_dimension(self) -> int:
        """int: The input vector dimensionality."""
        return self.embedding_layer[0].in_features

    @property
    def embedding_dimension(self) -> int:
        """int: The embedding dimension for each vector."""
        return self.embedding_layer[-1].out_features

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        batch_size = x.shape[0]
        v = x[:, self._embedded_indices].reshape(-1, self.vector_dimension)      # (batch * n_vectors, vector_dim)
        # The vector and its flip go through the networks as two separate calls of the same shape: then the result
        # for -x is bit-for-bit the result for x (the two terms of the sum just swap).
        flipped = -v
        candidates = torch.stack([self.embedding_layer(v), self.embedding_layer(flipped)], dim=1)        # (.., 2, E)
        weights = torch.softmax(torch.stack([self.weight_layer(v), self.weight_layer(flipped)], dim=1), dim=1)
        embedded = (weights * candidates).sum(dim=1)                              # (.., embedding_dim)
        return torch.cat([x[:, self._nonembedded_indices], embedded.reshape(batch_size, -1)], dim=1)

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        vec_degrees = degrees_in[self._embedded_indices].reshape(-1, self.vector_dimension)
        if not torch.all(vec_degrees == vec_degrees[:, [0]]):
            raise ValueError('The same degree must be assigned to all '
                             'components of each embedded vectors.')
        return torch.cat([degrees_in[self._nonembedded_indices],
                          vec_degrees[:, [0]].expand(-1, self.embedding_dimension).flatten()])


class MixedEmbedding(MAFEmbedding):
    """Several embeddings side by side, each on its own features (reference mafembed.py:354-446): output =
    ``[non-embedded features, layer 0 output, layer 1 output, ...]``."""

    def __init__(self, n_features_in: int, embedding_layers: Sequence[MAFEmbedding],
                 embedded_indices: Sequence[Sequence[int]]):
        super().__init__()
        if len(embedding_layers) != len(embedded_indices):
            raise ValueError('Different number of layers and indices.')
        embedded_indices = [ensure_tensor_sequence(indices) for indices in embedded_indices]
        first = set(embedded_indices[0].tolist())
        for indices in embedded_indices[1:]:
            if len(first & set(indices.tolist())) > 0:
                raise ValueError('Different embedding layers must be assigned '
                                 'to different feature indices.')
        self.embedding_layers = torch.nn.ModuleList(embedding_layers)
        for i, indices in enumerate(embedded_indices):
            self.register_buffer(f'_embedded_indices{i}', indices)
        self.register_buffer('_nonembedded_indices', remove_and_shift_sorted_indices(
            indices=torch.arange(n_features_in), removed_indices=torch.cat(embedded_indices).sort().values,
            shift=False))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        parts = [layer(x[:, getattr(self, f'_embedded_indices{i}')].contiguous())
                 for i, layer in enumerate(self.embedding_layers)]
        return torch.cat([x[:, self._nonembedded_indices], *parts], dim=1)

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        parts = [layer.get_degrees_out(degrees_in[getattr(self, f'_embedded_indices{i}')])
                 for i, layer in enumerate(self.embedding_layers)]
        return torch.cat([degrees_in[self._nonembedded_indices], *parts])
