"""Affine and volume-preserving transformers (reference ``tfep/nn/transformers/affine.py``)."""
from typing import Optional

import torch

from ... import ops, torch_ops  # noqa: F401  (torch_ops registers torch.ops.tfep.*)
from .transformer import MAFTransformer


class AffineTransformer(MAFTransformer):
    r""":math:`y_i = \exp(a_i) x_i + b_i` (reference affine.py:28-141).

    ``parameters[:, i]`` is the shift and ``parameters[:, n_features + i]`` the log scale of
    feature ``i`` (affine.py:136-141).
    """
    n_parameters_per_feature = 2

    def forward(self, x, parameters):
        ops.check_device_tensor(x, 'x')
        return tuple(torch.ops.tfep.affine_forward(x, parameters))        # differentiable (tfep::affine_backward)

    def inverse(self, y, parameters):
        ops.check_device_tensor(y, 'y')
        return tuple(torch.ops.tfep.affine_inverse(y, parameters))

    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        return torch.zeros(size=(self.n_parameters_per_feature * n_features,))

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return degrees_in.tile((self.n_parameters_per_feature,))


class VolumePreservingShiftTransformer(MAFTransformer):
    r""":math:`y_i = x_i + b_i` with optional periodic wrap, ``log_det_J = 0`` (reference affine.py:148-274)."""
    n_parameters_per_feature = 1

    def __init__(self, periodic_indices: Optional[torch.Tensor] = None,
                 periodic_limits: Optional[torch.Tensor] = None):
        super().__init__()
        self.periodic_indices = periodic_indices
        self.periodic_limits = periodic_limits

    def _mask(self, x):
        if self.periodic_indices is None:
            return None, (0.0, 1.0)
        m = torch.zeros(x.shape[1], dtype=torch.int32)
        m[torch.as_tensor(self.periodic_indices).long().cpu()] = 1
        lim = [float(v) for v in self.periodic_limits]
        return m.to(x.device), (lim[0], lim[1])

    def forward(self, x, parameters):
        m, lim = self._mask(x)
        return ops.volume_preserving_shift(x, parameters, m, lim, inverse=False)

    def inverse(self, y, parameters):
        m, lim = self._mask(y)
        return ops.volume_preserving_shift(y, parameters, m, lim, inverse=True)

    def get_identity_parameters(self, n_features: int) -> torch.Tensor:
        return torch.zeros(size=(self.n_parameters_per_feature * n_features,))

    def get_degrees_out(self, degrees_in: torch.Tensor) -> torch.Tensor:
        return degrees_in.tile((self.n_parameters_per_feature,))
