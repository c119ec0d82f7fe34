"""Transformer protocols (reference ``tfep/nn/transformers/transformer.py:26-127``)."""
import abc

import torch


class Transformer(abc.ABC, torch.nn.Module):
    """``forward(x, parameters) -> (y, log_det_J)``, ``inverse(y, parameters) -> (x, log_det_J)``."""

    def forward(self, x: torch.Tensor, parameters: torch.Tensor):
        return super().forward(x)  # Raises NotImplementedError.

    @abc.abstractmethod
    def inverse(self, y: torch.Tensor, parameters: torch.Tensor):
        pass

    @abc.abstractmethod
    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        """Parameters ``(n_parameters,)`` that make the transformer the identity."""
        pass


class MAFTransformer(Transformer):
    """A transformer usable in :class:`tfep_amd.nn.flows.MAF`."""

    @abc.abstractmethod
    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        """Autoregressive degrees of the conditioner outputs feeding this transformer."""
        pass
