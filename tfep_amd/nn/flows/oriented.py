"""Flow wrapper that constrains the rotational degrees of freedom (reference ``tfep/nn/flows/oriented.py:35-232``).

The change of frame is a per-sample 3x3 rotation (torch ops, ``tfep_amd.utils.geometry``); the wrapped flow
and :class:`PartialFlow`'s column gather / scatter run on the HIP kernels.
"""
import torch

from ...utils.geometry import batchwise_rotate, get_axis_from_name, reference_frame_rotation_matrix
from ...utils.misc import atom_to_flattened, flattened_to_atom
from .partial import PartialFlow

_AXES = 'xyz'


class OrientedFlow(PartialFlow):
    """Rotate each sample so that one point lies on ``axis`` and another on ``plane``, map the remaining
    3N-3 coordinates with the wrapped flow, and (optionally) rotate back.  3D only.

    Arguments (reference oriented.py:61-103): ``flow``; ``axis_point_idx`` / ``plane_point_idx`` (point, not
    feature, indices; default: the first two points, whichever is free); ``axis`` in 'x', 'y', 'z'; ``plane`` in
    'xy', 'yz', 'xz' containing the axis; ``round_off_imprecisions`` (set the constrained coordinates to exactly 0
    before the wrapped flow); ``rotate_back`` (required for ``inverse``); ``return_partial``.
    """

    def __init__(self, flow, axis_point_idx=None, plane_point_idx=None, axis='x', plane='xy',
                 round_off_imprecisions=True, rotate_back=True, return_partial=False):
        if return_partial and rotate_back:
            raise ValueError("'return_partial=True' is supported only if 'rotate_back=False'")
        if axis_point_idx is None:
            axis_point_idx = 1 if plane_point_idx == 0 else 0
        if plane_point_idx is None:
            plane_point_idx = 1 if axis_point_idx == 0 else 0
        if axis_point_idx == plane_point_idx:
            raise ValueError("'axis_point_idx' and 'plane_point_idx' must be different.")
        if axis not in plane:
            raise ValueError("To constrain 'plane_atom_idx' to stay on plane {plane} "
                             "'axis_atom_idx' must be constrained on an axis on the same plane.")

        # The frame in letters: `axis`, the other axis of the plane, and the axis normal to the plane.
        in_plane = next(n for n in plane if n != axis)
        normal = next(n for n in _AXES if n not in plane)
        e_axis, e_plane = get_axis_from_name(axis), get_axis_from_name(in_plane)

        # Constrained (zero) coordinates: the axis point keeps only its `axis` coordinate, the plane point loses
        # the one normal to the plane -- 3 DOFs in all, removed from the wrapped flow's input.
        zeroed = [3 * axis_point_idx + _AXES.index(n) for n in _AXES if n != axis]
        zeroed.append(3 * plane_point_idx + _AXES.index(normal))
        super().__init__(flow, fixed_indices=torch.tensor(zeroed), return_partial=return_partial)

        self.round_off_imprecisions = round_off_imprecisions
        self.rotate_back = rotate_back
        self._points = (int(axis_point_idx), int(plane_point_idx))          # host copies
        for name, value in (('_axis', e_axis), ('_plane_axis', e_plane),
                            ('_plane_normal', torch.linalg.cross(e_axis, e_plane)),      # signed: +-normal
                            ('_axis_point_idx', torch.as_tensor(axis_point_idx)),
                            ('_plane_point_idx', torch.as_tensor(plane_point_idx))):
            self.register_buffer(name, value)

    def forward(self, x):
        return self._in_frame(x, PartialFlow.forward)

    def inverse(self, y):
        if not self.rotate_back:
            raise ValueError("The inverse of OrientedFlow can be computed only"
                             " if 'rotate_back' is set to True during both the"
                             " forward and inverse transformations.")
        return self._in_frame(y, PartialFlow.inverse)

    def _in_frame(self, x, partial_pass):
        """Rotate into the constrained frame, run ``partial_pass`` (PartialFlow.forward / .inverse), rotate back."""
        pts = flattened_to_atom(x)
        a, p = self._points
        # nearest half-axis (not the positive one): the map stays invertible when the axis point flips
        rot = reference_frame_rotation_matrix(pts[:, a], pts[:, p], self._axis, self._plane_axis, self._plane_normal,
                                              project_on_positive_axis=False)
        framed = atom_to_flattened(batchwise_rotate(pts, rot))
        if self.round_off_imprecisions:
            framed = framed.index_fill(1, self._fixed_indices, 0.0)
        out = partial_pass(self, framed)
        if self.return_partial or not self.rotate_back:
            return out
        y = atom_to_flattened(batchwise_rotate(flattened_to_atom(out[0]), rot, inverse=True))
        return (y, *out[1:])
