"""Autoregressive flow (reference ``tfep/nn/flows/autoregressive.py:29-247``).

Same constructor, buffers (``_transformer_indices``, ``_inverse_masks``, ``_fixed_indices``,
``_conditioner_indices``) and semantics.  Two execution paths, both on the HIP kernels:

* generic: ``parameters = conditioner(x)`` then ``transformer(x, parameters)`` -- works with any
  user-supplied conditioner / transformer module (the parameter tensor goes through HBM);
* fused (MADE conditioner + affine / supported spline transformer): the output layer of MADE
  and the transformer run in ONE kernel (``tfep_fused_output_transformer_forward``), the
  ``(batch, P*D)`` parameter tensor lives only in MFMA accumulators.
"""
import ctypes
from typing import Optional, Sequence

import torch

from ... import _lib, ops
from ...utils.misc import ensure_tensor_sequence
from ..conditioners.made import MADE
from ..transformers.affine import AffineTransformer
from ..transformers.spline import NeuralSplineTransformer

_FUSED_AFFINE, _FUSED_SPLINE = 0, 1


class AutoregressiveFlow(torch.nn.Module):
    """Transform features with a transformer parametrised by a conditioner (Papamakarios et al. 2017)."""

    def __init__(
            self,
            n_features_in: int,
            transformer_indices: Sequence[Sequence[int]],
            conditioner: torch.nn.Module,
            transformer: torch.nn.Module,
            conditioner_indices: Optional[Sequence[int]] = None,
            initialize_identity: bool = True,
    ):
        super().__init__()
        transformer_indices = [ensure_tensor_sequence(x) for x in transformer_indices]
        # An empty buffer stands for None (reference autoregressive.py:93-98).
        if conditioner_indices is None:
            conditioner_indices = torch.tensor([], dtype=int)
        else:
            conditioner_indices = ensure_tensor_sequence(conditioner_indices)

        for indices in (conditioner_indices, *transformer_indices):
            if (indices is not None) and torch.any((indices < 0) | (n_features_in <= indices)):
                raise ValueError("All indices must be 0 <= i < n_features_in.")

        n_iter = len(transformer_indices)
        inverse_masks = torch.full((n_iter, n_features_in), False)
        for idx, indices in enumerate(transformer_indices):
            inverse_masks[idx, indices] = True

        transformer_indices = torch.cat(transformer_indices).sort().values
        fixed_indices = torch.arange(n_features_in)
        fixed_indices = fixed_indices[~torch.isin(fixed_indices, transformer_indices)]
        n_transformer_indices = len(transformer_indices)
        if len(fixed_indices) == 0:
            transformer_indices = torch.empty_like(fixed_indices)

        self._conditioner = conditioner
        self._transformer = transformer
        self.register_buffer('_transformer_indices', transformer_indices)
        self.register_buffer('_inverse_masks', inverse_masks)
        self.register_buffer('_fixed_indices', fixed_indices)
        self.register_buffer('_conditioner_indices', conditioner_indices)
        self._dev = {}
        self.fused = True      # set False to force the generic (unfused) path

        if initialize_identity:
            identity_parameters = self._transformer.get_identity_parameters(n_transformer_indices)
            self._conditioner.set_output(identity_parameters)

    @property
    def has_fixed_indices(self):
        """bool: True if some of the features are not transformed by the flow."""
        return len(self._fixed_indices) > 0

    def _apply(self, fn, *args, **kwargs):
        self._dev = {}
        return super()._apply(fn, *args, **kwargs)

    # ------------------------------------------------------------------ device-side index tables
    def _tables(self, device):
        key = str(device)
        t = self._dev.get(key)
        if t is not None:
            return t
        i32 = dict(device=device, dtype=torch.int32)
        n_in = self._inverse_masks.shape[1]
        tr = self._transformer_indices.cpu() if self.has_fixed_indices else torch.arange(n_in, device='cpu')
        t = {
            'tr': tr.to(**i32), 'fixed': self._fixed_indices.to(**i32),
            'cond': self._conditioner_indices.to(**i32),
            'n_tr': len(tr),
        }
        self._dev[key] = t
        return t

    def _inverse_steps(self, device):
        """Per inverse pass: the columns of x to commit and their position among the transformer
        features (reference autoregressive.py:203-227)."""
        t = self._tables(device)
        if 'inverse_steps' not in t:
            i32 = dict(device=device, dtype=torch.int32)
            n_in = self._inverse_masks.shape[1]
            tr = t['tr'].cpu().long()
            pos = torch.full((n_in,), -1, dtype=torch.long, device='cpu')
            pos[tr] = torch.arange(len(tr), device='cpu')
            steps = []
            for m in self._inverse_masks.cpu():
                cols = torch.nonzero(m).flatten()
                steps.append((cols.to(**i32), pos[cols].to(**i32)))
            t['inverse_steps'] = steps
        return t['inverse_steps']

    # ------------------------------------------------------------------ fused path
    def _fused_kind(self):
        if not self.fused or not isinstance(self._conditioner, MADE) or len(self._conditioner_indices) > 0:
            return None
        tr = self._transformer
        if type(tr) is AffineTransformer:
            return _FUSED_AFFINE
        if type(tr) is NeuralSplineTransformer and int(tr.n_bins) == 8 and not bool(tr._identity_boundary_slopes) \
                and not bool(tr._learn_lower_bound) and not bool(tr._learn_upper_bound):
            return _FUSED_SPLINE
        return None

    def _fused_plan(self, device, kind, tables):
        key = ('fused', str(device), kind)
        fp = self._dev.get(key)
        if fp is not None:
            return fp
        lib = _lib.load()
        made = self._conditioner
        mplan = made.plan(device)
        last = made.layers[-1]
        n_tr = tables['n_tr']
        P = 2 if kind == _FUSED_AFFINE else 25
        if last.out_features != P * n_tr:
            raise ValueError('conditioner output does not match the transformer parameters')
        desc = self._transformer.config(device).desc if kind == _FUSED_SPLINE else None
        tile_cols = lib.tfep_fused_tile_columns(kind, ctypes.byref(desc) if desc is not None else None)
        FT = tile_cols // (16 * P)
        n_slots = ops.round_up(n_tr, 16 * FT)
        deg_tr = made._degrees[-1][:n_tr].cpu()
        order = torch.argsort(deg_tr, stable=True)                 # slot -> transformed feature
        slot_of = torch.empty_like(order)
        slot_of[order] = torch.arange(n_tr, device='cpu')
        feat_tr = torch.zeros(n_slots, dtype=torch.long, device='cpu')
        feat_tr[:n_tr] = order
        feat_index = torch.full((n_slots,), -1, dtype=torch.long, device='cpu')
        feat_index[:n_tr] = tables['tr'].cpu().long()[order]
        s = slot_of.repeat(P)                                      # slot of output row o = p*n_tr + t
        p = torch.arange(P, device='cpu').repeat_interleave(n_tr)
        row_of_out = (s // (16 * FT)) * tile_cols + (((s // 16) % FT) * P + p) * 16 + (s % 16)
        n_tiles = n_slots // (16 * FT)
        i32 = dict(device=device, dtype=torch.int32)
        li = len(mplan['n_pad']) - 1
        fp = {
            'kind': kind, 'P': P, 'FT': FT, 'n_slots': n_slots, 'n_rows': n_tiles * tile_cols,
            'row_of_out': row_of_out.to(**i32), 'feat_index': feat_index.to(**i32), 'feat_tr': feat_tr.to(**i32),
            'li': li,
        }
        fp['k_ranges'] = ops.mask_k_ranges(last.mask, tile_cols, n_tiles, mplan['k_pad'][li],
                                           fp['row_of_out'], mplan['col_of_in'][li])
        fp['tile_order'] = ops.heavy_first_order(fp['k_ranges'])
        self._dev[key] = fp
        return fp

    def _forward_fused(self, x, kind):
        x, ldx = _lib.rows(x, 'x')
        B, D = x.shape
        tables = self._tables(x.device)
        fp = self._fused_plan(x.device, kind, tables)
        made = self._conditioner
        h, mplan = made.forward_hidden(x)
        w, b = made._pack_layer(mplan, fp['li'], made.layers[-1], row_of_out=fp['row_of_out'], n_rows=fp['n_rows'])
        y = x.clone() if self.has_fixed_indices else torch.empty(B, D, dtype=x.dtype, device=x.device)
        ldj = torch.empty(B, dtype=torch.float32, device=x.device)
        ws = torch.empty(fp['n_slots'] // 16, B, dtype=torch.float64, device=x.device)
        desc = self._transformer.config(x.device).desc if kind == _FUSED_SPLINE else None
        prof = getattr(self, '_profile_events', None)
        if prof is not None:                     # bench.py: HIP events around the fused launch
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(x.device))
        _lib.call('tfep_fused_output_transformer_forward', _lib.ptr(h), h.shape[1], _lib.ptr(w), w.shape[1],
                  _lib.ptr(b), _lib.ptr(fp['k_ranges']), _lib.ptr(fp['tile_order']), kind,
                  ctypes.byref(desc) if desc is not None else None,
                  _lib.ptr(x), ldx, _lib.ptr(y), D, _lib.ptr(fp['feat_index']), _lib.ptr(fp['feat_tr']),
                  fp['n_slots'], _lib.ptr(ws), _lib.ptr(ldj), 0, B, fp['n_rows'], w.shape[1], _lib.stream_of(x))
        if prof is not None:
            ev1.record(torch.cuda.current_stream(x.device))
            prof.append((ev0, ev1))
        return y, ldj

    # ------------------------------------------------------------------ reference API
    def forward(self, x: torch.Tensor):
        """``(y, log_det_J)`` of the push-forward (reference autoregressive.py:144-177)."""
        ops.check_device_tensor(x, 'x')
        kind = self._fused_kind()
        if kind is not None:
            return self._forward_fused(x, kind)
        parameters = self.get_transformer_parameters(x)
        if self.has_fixed_indices:
            t = self._tables(x.device)
            y = x.clone()                               # fixed features propagate unchanged
            y_tr, log_det_J = self._transformer(ops.gather_columns(x, t['tr']), parameters)
            ops.scatter_columns(y_tr, t['tr'], y)
        else:
            y, log_det_J = self._transformer(x, parameters)
        return y, log_det_J

    def inverse(self, y: torch.Tensor):
        """``(x, log_det_J)`` of the inverse map: one conditioner pass per autoregressive degree
        (reference autoregressive.py:179-229); the last pass' log-det is the total."""
        ops.check_device_tensor(y, 'y')
        t = self._tables(y.device)
        x = torch.zeros(y.shape, dtype=y.dtype, device=y.device)
        if self.has_fixed_indices:
            ops.scatter_columns(ops.gather_columns(y, t['fixed']), t['fixed'], x)
            y = ops.gather_columns(y, t['tr'])
        log_det_J = None
        freeze = getattr(self._conditioner, 'frozen_weights', None)
        ctx = freeze() if freeze is not None else _null_context()
        with ctx:
            for cols, pos in self._inverse_steps(y.device):
                parameters = self.get_transformer_parameters(x)
                x_temp, log_det_J = self._transformer.inverse(y, parameters)
                ops.scatter_columns(ops.gather_columns(x_temp, pos), cols, x)
        return x, log_det_J

    def get_transformer_parameters(self, x: torch.Tensor) -> torch.Tensor:
        """Run the conditioner (reference autoregressive.py:231-247)."""
        if len(self._conditioner_indices) > 0:
            x = ops.gather_columns(x, self._tables(x.device)['cond'])
        return self._conditioner(x)


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
