"""Flow acting on a subset of the degrees of freedom (reference ``tfep/nn/flows/partial.py:29-121``).

``TFEPMapBase.create_partial_flow`` (reference app/base.py:573-599) wraps the MAF stack in this module when
some atoms are fixed.  Column gather / scatter run on the HIP kernels (``tfep_gather_columns`` /
``tfep_scatter_columns``) and are differentiable (each is the other's adjoint).
"""
from typing import Sequence, Tuple

import torch

from ... import ops
from ...utils.misc import ensure_tensor_sequence


class _GatherColumns(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        ctx.save_for_backward(idx)
        ctx.n = x.shape[1]
        return ops.gather_columns(x.detach(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        gx = torch.zeros(g.shape[0], ctx.n, dtype=g.dtype, device=g.device)
        ops.scatter_columns(g.contiguous(), idx, gx)
        return gx, None


class _ReplaceColumns(torch.autograd.Function):
    """``y = base`` with ``y[:, idx] = src`` (fresh tensor; inputs untouched)."""

    @staticmethod
    def forward(ctx, base, src, idx):
        ctx.save_for_backward(idx)
        y = base.detach().clone()
        ops.scatter_columns(src.detach(), idx, y)
        return y

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        g = g.contiguous()
        gsrc = ops.gather_columns(g, idx)
        gbase = g.clone()
        ops.scatter_columns(torch.zeros_like(gsrc), idx, gbase)
        return gbase, gsrc, None


class PartialFlow(torch.nn.Module):
    """Map only the non-fixed degrees of freedom with the wrapped flow; the fixed ones are constants that
    the wrapped flow never sees.  Arguments and attributes as reference partial.py:57-68."""

    def __init__(self, flow: torch.nn.Module, fixed_indices: Sequence[int], return_partial: bool = False):
        super().__init__()
        self.flow = flow
        self.return_partial = return_partial
        self.register_buffer('_fixed_indices', ensure_tensor_sequence(fixed_indices))
        self.register_buffer('_propagated_indices', None)
        self._i32 = {}

    def _apply(self, fn, *args, **kwargs):
        self._i32 = {}
        return super()._apply(fn, *args, **kwargs)

    def n_parameters(self):
        """int: The total number of parameters that can be optimized."""
        return self.flow.n_parameters()

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor]:
        return self._pass(x, inverse=False)

    def inverse(self, y: torch.Tensor) -> Tuple[torch.Tensor]:
        return self._pass(y, inverse=True)

    def _indices(self, x):
        key = (str(x.device), x.shape[1])
        if key not in self._i32:
            fixed = set(self._fixed_indices.tolist())
            prop = torch.tensor([i for i in range(x.shape[1]) if i not in fixed], dtype=torch.long)
            if self._propagated_indices is None:
                self._propagated_indices = prop.to(self._fixed_indices.device)
            self._i32[key] = prop.to(device=x.device, dtype=torch.int32)
        return self._i32[key]

    def _pass(self, x, inverse):
        has_fixed = len(self._fixed_indices) > 0
        x_in = x
        if has_fixed:
            ops.check_device_tensor(x, 'x')
            prop = self._indices(x)
            x_in = _GatherColumns.apply(x, prop)
        out = self.flow.inverse(x_in) if inverse else self.flow(x_in)
        if self.return_partial:
            return out
        if has_fixed:
            y = _ReplaceColumns.apply(x, out[0], prop)
        else:
            y = out[0]
        return (y, *out[1:])
