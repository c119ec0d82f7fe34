"""Rigid-frame helpers for the flow wrappers (reference ``tfep/utils/geometry.py:185-411``).

Only what ``OrientedFlow`` / ``CenteredCentroidFlow`` need.  The reference composes two Rodrigues
rotations from angles (acos / asin / sin / cos); here both rotations are written in closed form from dot
and cross products, which gives the same matrices without the inverse-trig round trip and stays
differentiable.  These are O(batch) 3x3 operations around the hot path and run as
ordinary torch ops on the input's device.
"""
from typing import Optional

import torch


def get_axis_from_name(name: str) -> torch.Tensor:
    """Unit vector of the named Cartesian axis (reference geometry.py:279-293)."""
    try:
        i = 'xyz'.index(name)
    except ValueError:
        raise ValueError("'name' must be one of 'x', 'y', 'z'") from None
    v = torch.zeros(3)
    v[i] = 1.0
    return v


def _skew(w):
    """(B,3) -> (B,3,3) cross-product matrices [w]x."""
    z = torch.zeros_like(w[:, 0])
    return torch.stack([
        torch.stack([z, -w[:, 2], w[:, 1]], dim=1),
        torch.stack([w[:, 2], z, -w[:, 0]], dim=1),
        torch.stack([-w[:, 1], w[:, 0], z], dim=1),
    ], dim=1)


def batchwise_rotate(x: torch.Tensor, rotation_matrices: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Rotate every point of ``x`` (B,N,3) by its sample's matrix (B,3,3) (reference geometry.py:239-276)."""
    if inverse:
        return torch.matmul(x, rotation_matrices)
    return torch.matmul(x, rotation_matrices.transpose(1, 2))


def reference_frame_rotation_matrix(
        axis_atom_positions: torch.Tensor,
        plane_atom_positions: torch.Tensor,
        axis: torch.Tensor,
        plane_axis: torch.Tensor,
        plane_normal: Optional[torch.Tensor] = None,
        project_on_positive_axis: bool = False,
) -> torch.Tensor:
    """Rotation matrices (B,3,3) that put one point on ``axis`` and another on the ``axis``/``plane_axis``
    plane (same contract as reference geometry.py:296-411).

    The first rotation is the minimal one taking the axis point's direction onto ``axis`` (or onto
    ``-axis`` when that is closer and ``project_on_positive_axis`` is False).  The second is about ``axis``
    and brings the plane point onto the nearest half of the ``plane_axis`` line, so the axis point stays put.
    """
    dtype = axis_atom_positions.dtype
    axis = axis.to(axis_atom_positions)
    plane_axis = plane_axis.to(axis_atom_positions)
    if plane_normal is None:
        plane_normal = torch.linalg.cross(axis, plane_axis, dim=0)
    plane_normal = plane_normal.to(axis_atom_positions)
    plane_normal = plane_normal / torch.linalg.vector_norm(plane_normal)
    eye = torch.eye(3, dtype=dtype, device=axis.device)

    # R1: minimal rotation u -> axis, with u = +-(unit axis point).
    u = axis_atom_positions / torch.linalg.vector_norm(axis_atom_positions, dim=1, keepdim=True)
    c = u @ axis
    if not project_on_positive_axis:
        flip = torch.where(c < 0, -torch.ones_like(c), torch.ones_like(c))
        u = u * flip.unsqueeze(1)
        c = c * flip
    w = torch.linalg.cross(u, axis.expand_as(u), dim=1)
    K = _skew(w)
    # Rodrigues with sin = |w|, cos = c:  I + [w]x + [w]x^2 / (1 + c).  For the antiparallel case
    # (only reachable with project_on_positive_axis=True) rotate by pi about plane_axis x axis.
    denom = 1.0 + c
    safe = denom > 1e-12
    r1 = eye + K + torch.matmul(K, K) / torch.where(safe, denom, torch.ones_like(denom))[:, None, None]
    if project_on_positive_axis:
        n = torch.linalg.cross(plane_axis, axis, dim=0)
        flip_pi = 2.0 * torch.outer(n, n) - eye
        r1 = torch.where(safe[:, None, None], r1, flip_pi)

    # R2: rotation about axis by phi with cos(phi) = |q_p|/|q|, sin(phi) = -sign(q_p) q_n/|q|, where
    # q is the rotated plane point projected perpendicular to axis.
    p = torch.matmul(r1, plane_atom_positions.unsqueeze(2)).squeeze(2)
    q_p = p @ plane_axis
    q_n = p @ plane_normal
    q_norm = torch.sqrt(q_p * q_p + q_n * q_n)
    sgn = torch.sign(q_p)
    cos2 = torch.where(sgn == 0, torch.ones_like(q_p), q_p.abs() / q_norm)
    sin2 = torch.where(sgn == 0, torch.zeros_like(q_p), -sgn * q_n / q_norm)
    aa = torch.outer(axis, axis)
    kx = _skew(axis.unsqueeze(0))[0]
    r2 = cos2[:, None, None] * eye + sin2[:, None, None] * kx + (1.0 - cos2)[:, None, None] * aa

    return torch.matmul(r2, r1)
