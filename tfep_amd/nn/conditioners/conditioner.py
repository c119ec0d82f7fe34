"""Conditioner protocol of an autoregressive flow (reference ``tfep/nn/conditioners/conditioner.py:26-63``)."""
import abc

import torch


class Conditioner(abc.ABC, torch.nn.Module):
    """A conditioner maps ``x (batch, n_features)`` to transformer parameters ``(batch, n_parameters)``."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return super().forward(x)  # Raises NotImplementedError.

    @abc.abstractmethod
    def set_output(self, output: torch.Tensor):
        """Make the conditioner produce the constant ``output`` (identity initialisation)."""
        pass
