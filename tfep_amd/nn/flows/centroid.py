"""Flow wrapper that constrains the (weighted) centroid (reference ``tfep/nn/flows/centroid.py:30-268``).

``TFEPMapBase`` puts this around the MAF stack to remove the translational degrees of freedom.  The
translation and centroid arithmetic are O(batch x features) torch ops on the input's device; the wrapped
flow and the column gather / scatter of :class:`PartialFlow` run on the HIP kernels.
"""
from typing import Optional, Sequence, Tuple

import torch

from ...utils.misc import atom_to_flattened, atom_to_flattened_indices, ensure_tensor_sequence, flattened_to_atom
from .partial import PartialFlow


def _optional_tensor(values):
    return None if values is None else ensure_tensor_sequence(values)


class CenteredCentroidFlow(PartialFlow):
    """Translate the centroid to ``origin``, map all points but one with the wrapped flow, then place the
    remaining point so that the centroid is unchanged.  Arguments and attributes as reference
    centroid.py:52-112."""

    def __init__(
            self,
            flow: torch.nn.Module,
            space_dimension: int,
            subset_point_indices: Optional[Sequence[int]] = None,
            weights: Optional[Sequence[float]] = None,
            fixed_point_idx: int = 0,
            origin: Optional[Sequence[float]] = None,
            translate_back: bool = True,
            return_partial: bool = False,
    ):
        if translate_back and return_partial:
            raise ValueError("'return_partial=True' is supported only if 'translate_back=False'")
        if origin is not None and len(origin) != space_dimension:
            raise ValueError("'origin' must have length equal to 'space_dimension'.")
        subset, weights = _optional_tensor(subset_point_indices), _optional_tensor(weights)
        if subset is not None and weights is not None and len(weights) != len(subset):
            raise ValueError("'weights' must have the same length as 'subset_point_indices'.")

        # `fixed_point_idx` counts within the subset when there is one: the point itself is subset[fixed_point_idx].
        fixed_point = int(fixed_point_idx if subset is None else subset[fixed_point_idx])
        super().__init__(flow, fixed_indices=atom_to_flattened_indices(torch.tensor([fixed_point]), space_dimension),
                         return_partial=return_partial)

        self._space_dimension = space_dimension
        self.translate_back = translate_back
        self.register_buffer('_fixed_point_idx', torch.as_tensor(fixed_point_idx))
        self.register_buffer('_subset_point_indices', subset)
        # normalised, as a column: multiplies (batch, n_points, dim) directly
        self.register_buffer('_weights', None if weights is None else (weights / weights.sum()).unsqueeze(1))
        self.register_buffer('origin', torch.zeros(space_dimension) if origin is None else ensure_tensor_sequence(origin))
        # Host copies so that no pass reads a device scalar back.
        self._host_fixed_point_idx = int(fixed_point_idx)
        self._host_fixed_point = fixed_point
        self._single_point_centroid = subset is not None and len(subset) <= 1

    @property
    def space_dimension(self):
        """int: The dimensionality of a single point in space."""
        return self._space_dimension

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        return self._transform(x, inverse=False)

    def inverse(self, y: torch.Tensor) -> Tuple[torch.Tensor]:
        if not self.translate_back:
            raise ValueError("The inverse of CenteredCentroidFlow can be computed"
                             " only if 'translate_back' is set to True during both"
                             " the forward and inverse transformations.")
        return self._transform(y, inverse=True)

    def _transform(self, x, inverse):
        dim = self._space_dimension
        pts = flattened_to_atom(x, dim)
        shift = (self.origin - self._centroid(pts)).unsqueeze(1)
        x_centered = atom_to_flattened(pts + shift)

        out = PartialFlow.inverse(self, x_centered) if inverse else PartialFlow.forward(self, x_centered)
        if self.return_partial:
            return out
        y = out[0]

        # Put the fixed point where it restores the centroid (nothing to do when the centroid IS that point).
        if not self._single_point_centroid:
            y_pts = flattened_to_atom(y, dim)
            rest, w_fixed = self._centroid(y_pts, exclude_fixed_point=True)
            fixed_pos = (self.origin - rest) / w_fixed
            f = self._host_fixed_point
            y_pts = torch.cat([y_pts[:, :f], fixed_pos.unsqueeze(1), y_pts[:, f + 1:]], dim=1)
            y = atom_to_flattened(y_pts)
        if self.translate_back:
            y = atom_to_flattened(flattened_to_atom(y, dim) - shift)
        return (y, *out[1:])

    def _centroid(self, pts, exclude_fixed_point=False):
        """Centroid (B, dim) of the selected points; with ``exclude_fixed_point`` the fixed point's
        contribution is left out and its weight is returned too."""
        if self._subset_point_indices is not None:
            pts = pts[:, self._subset_point_indices]
        if self._weights is None:
            centroid = pts.mean(dim=1)
            w_fixed = 1.0 / pts.shape[1]
        else:
            centroid = (pts * self._weights).sum(dim=1)
            w_fixed = self._weights[self._host_fixed_point_idx]
        if exclude_fixed_point:
            return centroid - pts[:, self._host_fixed_point_idx] * w_fixed, w_fixed
        return centroid
