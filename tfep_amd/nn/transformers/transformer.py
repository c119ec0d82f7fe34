"""Transformer protocols (reference ``tfep/nn/transformers/transformer.py:26-127``)."""
import abc

import torch


class Transformer(torch.nn.Module, metaclass=abc.ABCMeta):
    """Element-wise invertible map driven by conditioner parameters:
    ``forward(x, parameters) -> (y, log_det_J)`` and ``inverse(y, parameters) -> (x, log_det_J)``."""

    @abc.abstractmethod
    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        """Parameters ``(n_parameters,)`` for which the transformer is the identity map."""

    @abc.abstractmethod
    def inverse(self, y: torch.Tensor, parameters: torch.Tensor):
        """``(x, log_det_J)`` of the inverse map."""

    def forward(self, x: torch.Tensor, parameters: torch.Tensor):
        raise NotImplementedError(f'{type(self).__name__} must implement forward(x, parameters)')


class MAFTransformer(Transformer):
    """A transformer that can sit in a :class:`tfep_amd.nn.flows.MAF`: it also tells the MADE conditioner which
    autoregressive degree each of its parameters belongs to."""

    @abc.abstractmethod
    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        """Degrees of the conditioner outputs (one per parameter) given the degrees of the transformed features."""
