"""Flow wrapper that constrains the rotational degrees of freedom (reference ``tfep/nn/flows/oriented.py:35-232``).

The change of frame is a per-sample 3x3 rotation (torch ops, ``tfep_amd.utils.geometry``); the wrapped flow
and :class:`PartialFlow`'s column gather / scatter run on the HIP kernels.
"""
from typing import Optional, Tuple

import torch

from ...utils.geometry import batchwise_rotate, get_axis_from_name, reference_frame_rotation_matrix
from ...utils.misc import atom_to_flattened, atom_to_flattened_indices, flattened_to_atom
from .partial import PartialFlow


class OrientedFlow(PartialFlow):
    """Rotate each sample so that one point lies on ``axis`` and another on ``plane``, map the remaining
    3N-3 coordinates with the wrapped flow, and (optionally) rotate back.  3D only.  Arguments as reference
    oriented.py:61-103."""

    def __init__(
            self,
            flow: torch.nn.Module,
            axis_point_idx: Optional[int] = None,
            plane_point_idx: Optional[int] = None,
            axis: str = 'x',
            plane: str = 'xy',
            round_off_imprecisions: bool = True,
            rotate_back: bool = True,
            return_partial: bool = False,
    ):
        if return_partial and rotate_back:
            raise ValueError("'return_partial=True' is supported only if 'rotate_back=False'")
        # Default points: the first two, whichever is not taken.
        if axis_point_idx is None:
            axis_point_idx = 0 if plane_point_idx != 0 else 1
        if plane_point_idx is None:
            plane_point_idx = 0 if axis_point_idx != 0 else 1
        if axis_point_idx == plane_point_idx:
            raise ValueError("'axis_point_idx' and 'plane_point_idx' must be different.")
        if axis not in plane:
            raise ValueError("To constrain 'plane_atom_idx' to stay on plane {plane} "
                             "'axis_atom_idx' must be constrained on an axis on the same plane.")

        axis_vector = get_axis_from_name(axis)
        plane_axis_vector = get_axis_from_name([n for n in 'xyz' if n != axis and n in plane][0])
        plane_normal_vector = torch.linalg.cross(axis_vector, plane_axis_vector)

        # 2 DOFs of the axis point (those off the axis) and 1 DOF of the plane point (off the plane) are fixed at 0.
        axis_dofs = atom_to_flattened_indices(torch.tensor([axis_point_idx]))
        plane_dofs = atom_to_flattened_indices(torch.tensor([plane_point_idx]))
        fixed_indices = torch.cat([axis_dofs[axis_vector == 0.0], plane_dofs[plane_normal_vector != 0.0]])
        super().__init__(flow, fixed_indices=fixed_indices, return_partial=return_partial)

        self.register_buffer('_axis', axis_vector)
        self.register_buffer('_plane_axis', plane_axis_vector)
        self.register_buffer('_plane_normal', plane_normal_vector)
        self.register_buffer('_axis_point_idx', torch.as_tensor(axis_point_idx))
        self.register_buffer('_plane_point_idx', torch.as_tensor(plane_point_idx))
        self.round_off_imprecisions = round_off_imprecisions
        self.rotate_back = rotate_back
        self._host_axis_point = int(axis_point_idx)
        self._host_plane_point = int(plane_point_idx)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        return self._transform(x)

    def inverse(self, y: torch.Tensor) -> Tuple[torch.Tensor]:
        if not self.rotate_back:
            raise ValueError("The inverse of OrientedFlow can be computed only"
                             " if 'rotate_back' is set to True during both the"
                             " forward and inverse transformations.")
        return self._transform(y, inverse=True)

    def _transform(self, x, inverse=False):
        pts = flattened_to_atom(x)
        rotation_matrices = reference_frame_rotation_matrix(
            axis_atom_positions=pts[:, self._host_axis_point],
            plane_atom_positions=pts[:, self._host_plane_point],
            axis=self._axis,
            plane_axis=self._plane_axis,
            plane_normal=self._plane_normal,
            # Nearest half-axis, so that the transformation stays invertible when the axis point flips.
            project_on_positive_axis=False,
        )
        x_rot = atom_to_flattened(batchwise_rotate(pts, rotation_matrices))
        if self.round_off_imprecisions:
            x_rot = x_rot.index_fill(1, self._fixed_indices, 0.0)

        out = PartialFlow.inverse(self, x_rot) if inverse else PartialFlow.forward(self, x_rot)
        if self.return_partial:
            return out
        y = out[0]
        if self.rotate_back:
            y = atom_to_flattened(batchwise_rotate(flattened_to_atom(y), rotation_matrices, inverse=True))
        return (y, *out[1:])
