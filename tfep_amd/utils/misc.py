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
