"""Conditioner protocol of an autoregressive flow (reference ``tfep/nn/conditioners/conditioner.py:26-63``)."""
import abc

import torch


class Conditioner(torch.nn.Module, metaclass=abc.ABCMeta):
    """Maps ``x (batch, n_features)`` to the transformer parameters ``(batch, n_parameters)``.

    Subclasses provide ``forward`` and ``set_output``; :class:`tfep_amd.nn.conditioners.MADE` is the one with HIP
    kernels behind it, any other differentiable module works on the generic path.
    """

    @abc.abstractmethod
    def set_output(self, output: torch.Tensor):
        """Make the conditioner return the constant ``output`` (used for the identity initialisation)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError(f'{type(self).__name__} must implement forward(x)')
