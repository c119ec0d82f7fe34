"""Moebius transformer (reference ``tfep/nn/transformers/moebius.py:27-190``)."""
import torch

from ... import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.tfep.*)
from .transformer import MAFTransformer


class MoebiusTransformer(MAFTransformer):
    r""":math:`y = \frac{\|x\|^2 - \|w\|^2}{\|x - w\|^2}(x - w) - w` on ``dimension``-vectors.

    ``w`` is rescaled to ``max_radius/(1+|w|) * |x| * w`` (``|x| = 1`` if ``unit_sphere``); the
    inverse is the forward map with ``-w`` (reference moebius.py:142-147).
    """

    def __init__(self, dimension: int, max_radius: float = 0.99, unit_sphere: bool = False):
        super().__init__()
        self.dimension = dimension
        self.max_radius = max_radius
        self.unit_sphere = unit_sphere

    def forward(self, x, parameters):
        ops.check_device_tensor(x, 'x')
        return tuple(torch.ops.tfep.moebius_forward(x, parameters, int(self.dimension), float(self.max_radius),
                                                    bool(self.unit_sphere)))                   # differentiable

    def inverse(self, y, parameters):
        ops.check_device_tensor(y, 'y')
        return tuple(torch.ops.tfep.moebius_inverse(y, parameters, int(self.dimension), float(self.max_radius),
                                                    bool(self.unit_sphere)))

    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        return torch.zeros(size=(n_features,))

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return degrees_in.detach().clone()
