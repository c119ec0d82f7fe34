"""Host-side helpers shared by the modules (reference ``tfep/utils/misc.py``)."""
import numpy as np
import torch


def ensure_tensor_sequence(x, dtype=None):
    """Sequences become tensors, scalars / strings pass through (reference utils/misc.py:158-180)."""
    if not np.isscalar(x):
        try:
            x = torch.as_tensor(x, dtype=dtype)
        except (TypeError, RuntimeError):
            pass
    return x


def remove_and_shift_sorted_indices(indices, removed_indices, remove=True, shift=True):
    """Remove ``removed_indices`` from sorted ``indices`` and optionally shift the rest down
    (reference utils/misc.py:262-352; only the behaviours used on the flow path)."""
    indices = torch.as_tensor(indices)
    removed_indices = torch.as_tensor(removed_indices)
    if shift:
        shifts = torch.searchsorted(removed_indices, indices)
    if remove:
        keep = ~torch.isin(indices, removed_indices)
        indices = indices[keep]
        if shift:
            shifts = shifts[keep]
    if shift:
        indices = indices - shifts
    return indices


def flattened_to_atom(positions, space_dimension=3):
    """(B, n_atoms*dim) or (n_atoms*dim,) -> (B, n_atoms, dim) / (n_atoms, dim) view (reference utils/misc.py:28-59)."""
    n_atoms = positions.shape[-1] // space_dimension
    if positions.ndim > 1:
        return positions.reshape(positions.shape[0], n_atoms, space_dimension)
    return positions.reshape(n_atoms, space_dimension)


def atom_to_flattened(positions):
    """Inverse of :func:`flattened_to_atom` (reference utils/misc.py:62-91)."""
    n_atoms, dim = positions.shape[-2:]
    if positions.ndim > 2:
        return positions.reshape(positions.shape[0], n_atoms * dim)
    return positions.reshape(n_atoms * dim)


def atom_to_flattened_indices(atom_indices, space_dimension=3):
    """Point indices -> indices of their coordinates in the flattened layout (reference utils/misc.py:94-133):
    ``[0, 2] -> [0, 1, 2, 6, 7, 8]`` for ``space_dimension=3``."""
    if isinstance(atom_indices, torch.Tensor):
        offsets = torch.arange(space_dimension, dtype=atom_indices.dtype, device=atom_indices.device)
        return (atom_indices.unsqueeze(-1) * space_dimension + offsets).reshape(*atom_indices.shape[:-1], -1)
    atom_indices = np.asarray(atom_indices)
    return (atom_indices[..., None] * space_dimension + np.arange(space_dimension)).reshape(*atom_indices.shape[:-1], -1)
