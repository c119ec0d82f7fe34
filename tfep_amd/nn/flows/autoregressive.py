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
import math
import os
from typing import Optional, Sequence

import torch

from ... import _lib, ops
from ...utils.misc import ensure_tensor_sequence
from ..conditioners.made import MADE
from ..embeddings.mafembed import PeriodicEmbedding
from ..transformers.affine import AffineTransformer, VolumePreservingShiftTransformer
from ..transformers.mixed import MixedTransformer
from ..transformers.moebius import MoebiusTransformer
from ..transformers.spline import NeuralSplineTransformer
from .sequential import _side_stream

_FUSED_AFFINE, _FUSED_SPLINE, _FUSED_MIXED = 0, 1, 2      # (0 / 1: tfep_fused_kind; 2: one launch per group)


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
        # True: the fused output-GEMM + transformer kernel wherever one exists; False: never (the generic path: three
        # GEMMs + a transformer kernel); None: by size (_fused_pays)
        self.fused = None
        self.blocked_inverse = True   # set False to force the reference's one-full-pass-per-degree inverse
        self.split_gemm = None        # None: TFEP_SPLIT_GEMM (default on); False: exact-fp32 MFMA GEMMs in forward

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

    def _load_from_state_dict(self, *args, **kwargs):
        # the index tables, fused / blocked plans derive from the buffers a checkpoint replaces
        super()._load_from_state_dict(*args, **kwargs)
        self._dev = {}

    def _sync_conditioner(self):
        """After a ``load_state_dict`` the MADE conditioner re-derives its degrees from the loaded buffers; the input
        degrees come from this layer's own ``_inverse_masks`` (feature c is transformed at step d iff
        ``_inverse_masks[d, c]``; every other feature conditions: -1), mapped through the embedding if there is one."""
        made = self._conditioner
        if isinstance(made, MADE):
            made.begin_call()
        if not isinstance(made, MADE) or not made._degrees_stale:
            return
        self._dev = {}
        hint = None
        if len(self._conditioner_indices) == 0:
            deg_x = torch.full((self._inverse_masks.shape[1],), -1, dtype=torch.long)
            for d_, m_ in enumerate(self._inverse_masks.cpu()):
                deg_x[m_] = d_
            emb = getattr(made, 'embedding', None)
            hint = deg_x if emb is None else emb.get_degrees_out(deg_x.to(self._inverse_masks.device)).cpu()
        made._sync_degrees(hint)

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
    @staticmethod
    def _transformer_fused_kind(tr):
        """Fused epilogue of one transformer (the rule of tfep_fused_supported): affine; RQ splines of 8, 5 or 4 bins in
        every layout (at most 27 parameters per feature: 8 bins with both bounds learnable)."""
        if type(tr) is AffineTransformer:
            return _FUSED_AFFINE
        if type(tr) is NeuralSplineTransformer and tr.host()['n_bins'] in (4, 5, 8) and tr.n_parameters_per_feature <= 27:
            return _FUSED_SPLINE
        return None

    @staticmethod
    def _is_plain_shift(tr):
        """A VolumePreservingShiftTransformer without periodic features: y = x + b, log-det 0 (affine.py:366-456)."""
        return type(tr) is VolumePreservingShiftTransformer and tr.periodic_indices is None

    @classmethod
    def _member_fused_kind(cls, tr):
        """Fused epilogue of a member of a mixed transformer: as above, and the plain shift (MixedMAFMap's transformer of
        the reference-frame DOFs, app/mixedmaf.py:815-821) as an affine group whose log-scale rows stay zero."""
        return _FUSED_AFFINE if cls._is_plain_shift(tr) else cls._transformer_fused_kind(tr)

    def _fused_kind(self):
        if self.fused is False or not isinstance(self._conditioner, MADE) or len(self._conditioner_indices) > 0:
            return None
        tr = self._transformer
        if type(tr) is MixedTransformer:
            # every group on its own column tiles of the output GEMM, one fused launch per group
            if all(self._member_fused_kind(t) is not None for t in tr._transformers):
                return _FUSED_MIXED
            return None
        return self._transformer_fused_kind(tr)

    def prepack_async(self, device, stream, batch=None):
        """Start packing this layer's weights on ``stream`` for its next forward pass (called by SequentialFlow while
        the previous layer computes).  Only for the fused split-f16 path; a no-op otherwise."""
        kind = self._fused_kind()
        if kind is None or not self._use_split_gemm(batch) or not isinstance(self._conditioner, MADE):
            return
        # only worth the stream fork / join when the re-pack moves real data (cfg1-sized layers are launch bound)
        if not self._conditioner.split_worthwhile():
            return
        if torch.is_grad_enabled() and batch is not None and any(p.requires_grad for p in self._conditioner.parameters()):
            from . import _backward
            if _backward.saves_activations_at(self, batch):
                return                              # a training forward that keeps its activations packs the backward's way
        fp = self._fused_plan(device, kind, self._tables(device))
        self._conditioner.prepack_split_async(device, stream, last=(fp['row_of_out'], fp['n_rows']))

    def _use_split_gemm(self, batch=None):
        """Split-f16 GEMMs for the forward pass: ``self.split_gemm`` if set; else ``TFEP_SPLIT_GEMM`` (default on) when
        the conditioner is large enough for the operand conversions to pay (``MADE.split_worthwhile``: >= 4 M weights, or
        enough weight x row products at the given batch) -- smaller problems are launch bound and stay on the exact-fp32
        kernel."""
        if self.split_gemm is not None:
            return bool(self.split_gemm)
        if self._guard_exact:                      # this call's data failed the range guard (``_range_guard``)
            return False
        made = self._conditioner
        return ops.split_gemm_enabled() and isinstance(made, MADE) and made.split_worthwhile(batch)

    #: Data-aware guard of the split-f16 DEFAULT (``split_gemm = None``): ``None`` = on unless ``TFEP_SPLIT_GUARD=0``.
    #: The split format carries each activation row with ONE power-of-two scale: elements below 2^-19 of their row's maximum
    #: lose significance, and an output that sees only such elements (MADE's prefix masks make that possible) inherits the
    #: error component-wise (tests/test_gpu_split_gemm.py).  That is a property of FEATURES, not of single values (a lone
    #: small value in one row is harmless: 2 % of the rows of a 3000-feature Gaussian batch hold one, and every unit adds an
    #: exact fp32 bias and sees other features): before a forward that the size rule would send to the split kernels, the
    #: largest magnitude of every feature over the batch is taken, and if the non-zero ones span more than 2^19
    #: (``tfep_range_flag``) THIS call runs on the exact-fp32 MFMA kernels instead (``last_split_guard`` says so; one
    #: warning per layer).  Costs one read of x and one host synchronisation per layer call -- skipped inside a HIP-graph
    #: capture, where the host cannot wait (``graphs.GraphedFlow`` checks the same flag after every replay and repeats a
    #: flagged call eagerly), and when the arithmetic was chosen explicitly (``split_gemm = True / False``).
    #: ``inverse`` is guarded after the fact, on the x it has produced (``_inverse_impl``).
    #: The backward of a guarded forward stays exact too.
    #: Not guarded: the hidden activations (sums over many inputs plus an fp32 bias: no output sees "only small entries")
    #: and the weights (one scale per matrix from max |g|; an entry below 2^-19 of it contributes < 2^-19 max|w| |x| to a
    #: pre-activation that carries an exact fp32 bias) -- the wide-layer gradient goldens hold the default path to the
    #: reference's own float32 accuracy entry by entry (tests/test_gpu_backward.py).
    split_guard = None
    _guard_exact = False
    last_split_guard = None

    def _range_guard(self, x):
        """Context manager around one forward call: decides ``_guard_exact`` from the data (see ``split_guard``)."""
        return _RangeGuard(self, x)

    def _fused_plan(self, device, kind, tables):
        """Packed layout of the MADE output layer for the fused kernels.  The transformed features form GROUPS -- one for a
        plain transformer, one per sub-transformer of a MixedTransformer (whose parameters come grouped by transformer,
        mixed.py:64-68) -- and every group owns a run of column tiles (16 features x P parameters x FT) of ONE packed
        weight matrix: one re-pack per forward, one fused launch per group on its row slice."""
        key = ('fused', str(device), kind)
        fp = self._dev.get(key)
        if fp is not None:
            return fp
        lib = _lib.load()
        made = self._conditioner
        mplan = made.plan(device)
        last = made.layers[-1]
        n_tr = tables['n_tr']
        tr = self._transformer
        cols_tr = tables['tr'].cpu().long()                         # transformed feature -> column of x
        if kind == _FUSED_MIXED:
            members = [(self._member_fused_kind(t), t, ind.cpu().long(), off)
                       for t, ind, off in zip(tr._transformers, tr._indices, tr.host_splits()) if len(ind) > 0]
        else:
            members = [(kind, tr, None, 0)]
        i32 = dict(device=device, dtype=torch.int32)
        li = len(mplan['n_pad']) - 1
        row_of_out = torch.full((last.out_features,), -1, dtype=torch.long, device='cpu')
        groups, base = [], 0
        for k_g, t_g, rel, off in members:
            n_g = n_tr if rel is None else len(rel)
            P = 2 if k_g == _FUSED_AFFINE else t_g.n_parameters_per_feature      # parameter rows of the kernel's tile
            P_real = 1 if self._is_plain_shift(t_g) else P                       # rows the conditioner has (the shift:
            if off + P_real * n_g > last.out_features:                           # its log-scale rows stay zero)
                raise ValueError('conditioner output does not match the transformer parameters')
            desc = t_g.config(device).desc if k_g == _FUSED_SPLINE else None
            tile_cols = lib.tfep_fused_tile_columns(k_g, ctypes.byref(desc) if desc is not None else None)
            FT = tile_cols // (16 * P)
            n_slots = ops.round_up(n_g, 16 * FT)
            deg = made._degrees[-1][off:off + n_g].cpu()
            order = torch.argsort(deg, stable=True)                # slot -> feature of the group
            slot_of = torch.empty_like(order)
            slot_of[order] = torch.arange(n_g, device='cpu')
            feat_tr = torch.zeros(n_slots, dtype=torch.long, device='cpu')
            feat_tr[:n_g] = order
            feat_index = torch.full((n_slots,), -1, dtype=torch.long, device='cpu')
            feat_index[:n_g] = (cols_tr if rel is None else cols_tr[rel])[order]
            sl = slot_of.repeat(P_real)                            # slot of output row o = off + p*n_g + t
            pp = torch.arange(P_real, device='cpu').repeat_interleave(n_g)
            local = (sl // (16 * FT)) * tile_cols + (((sl // 16) % FT) * P + pp) * 16 + (sl % 16)
            n_tiles = n_slots // (16 * FT)
            row_of_out[off:off + P_real * n_g] = base + local
            grp = {'kind': k_g, 'transformer': t_g, 'P': P, 'FT': FT, 'n_slots': n_slots, 'n_rows': n_tiles * tile_cols,
                   'base': base, 'feat_index': feat_index.to(**i32), 'feat_tr': feat_tr.to(**i32)}
            grp['k_ranges'] = ops.mask_k_ranges(last.mask[off:off + P_real * n_g], tile_cols, n_tiles, mplan['k_pad'][li],
                                                local.to(**i32), mplan['col_of_in'][li])
            grp['tile_order'] = ops.heavy_first_order(grp['k_ranges'])
            groups.append(grp)
            base += grp['n_rows']
        if bool((row_of_out < 0).any()):
            raise ValueError('conditioner output does not match the transformer parameters')
        fp = {'kind': kind, 'groups': groups, 'n_rows': base, 'row_of_out': row_of_out.to(**i32), 'li': li}
        self._dev[key] = fp
        return fp

    def _forward_fused(self, x, kind):
        x, ldx = _lib.rows(x, 'x')
        B, D = x.shape
        tables = self._tables(x.device)
        fp = self._fused_plan(x.device, kind, tables)
        made = self._conditioner
        split = self._use_split_gemm(B)
        if split:
            h, h_inv, mplan = made.forward_hidden_split(x)
            w, w_inv, b, _ = made._pack_layer_split(mplan, fp['li'], made.layers[-1], row_of_out=fp['row_of_out'],
                                                    n_rows=fp['n_rows'])
        else:
            h, mplan = made.forward_hidden(x)
            w, b = made._pack_layer(mplan, fp['li'], made.layers[-1], row_of_out=fp['row_of_out'], n_rows=fp['n_rows'])
        prof = getattr(self, '_profile_events', None)
        if prof is not None:                     # bench.py: HIP events around the fused launch
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record(torch.cuda.current_stream(x.device))
        # torch.ops.tfep.fused_output_transformer = tfep_fused_output_transformer_forward[_split]; one launch per group
        # on the group's rows of the packed weights, every launch writing its own columns of y
        y = x.clone() if self.has_fixed_indices else torch.empty(x.shape, dtype=x.dtype, device=x.device)
        ldj = None
        for grp in fp['groups']:
            if grp['kind'] == _FUSED_SPLINE:
                cfg, hst = grp['transformer'].config(x.device), grp['transformer'].host()
                spl = (cfg.x0, cfg.xf, cfg.y0, cfg.yf, hst['n_bins'], hst['circular'], hst['identity'], hst['learn_lower'],
                       hst['learn_upper'], hst['min_bin'], hst['min_slope'])
            else:
                spl = (None, None, None, None, 0, False, False, False, False, 0.0, 0.0)
            r0, r1 = grp['base'], grp['base'] + grp['n_rows']
            l = torch.ops.tfep.fused_output_transformer_(
                h, h_inv if split else None, w[r0:r1], w_inv if split else None, b[r0:r1], grp['k_ranges'],
                grp['tile_order'], grp['kind'], x, y, grp['feat_index'], grp['feat_tr'], grp['n_slots'], grp['n_rows'], *spl)
            ldj = l if ldj is None else ldj + l
        if prof is not None:
            ev1.record(torch.cuda.current_stream(x.device))
            prof.append((ev0, ev1))
        if split:
            made.drop_packed_ahead()
        return y, ldj

    # ------------------------------------------------------------------ the whole layer in one kernel
    #: ``tfep_maf_layer_forward_split``: conditioner + transformer of the layer in ONE launch (csrc/maf_layer.hip) where that
    #: kernel exists -- a plain MADE (no embedding, no conditioning / fixed features) of 1 .. 3 hidden layers up to 4096 wide
    #: feeding a Moebius transformer of dimension 2, on the split-f16 arithmetic.  OPT-IN (True, or ``TFEP_LAYER_KERNEL=1``
    #: for every layer that qualifies): measured SLOWER than the launch-by-launch path at BASELINE cfg4-ii (11.5 against
    #: 10.6 ms; csrc/maf_layer.hip says why), so None / False leave it off.
    layer_kernel = None

    def _layer_kernel_ok(self, x):
        want = self.layer_kernel if self.layer_kernel is not None else os.environ.get('TFEP_LAYER_KERNEL', '0') == '1'
        if not want or self.fused is False:
            return False
        made, tr = self._conditioner, self._transformer
        if type(tr) is not MoebiusTransformer or tr.dimension != 2 or not isinstance(made, MADE):
            return False
        if getattr(made, 'embedding', None) is not None or self.has_fixed_indices or len(self._conditioner_indices) > 0:
            return False
        lins = made._linears()
        if not (2 <= len(lins) <= 4) or max(l.out_features for l in lins) > 4096 or lins[-1].out_features != x.shape[1]:
            return False
        if x.shape[1] % 2 or made.dimension_in != x.shape[1]:
            return False
        return True

    def _layer_plan(self, device):
        """k-ranges of every masked linear for the layer kernel's 256-column tiles (hidden units sorted by degree, the output
        layer in feature order), once per device and mask version."""
        made = self._conditioner
        mplan = made.plan(device)
        lp = mplan.get('layer_kernel')
        if lp is None:
            tn = _lib.load().tfep_maf_layer_tile_n()
            lins = made._linears()
            kr = []
            for li, lin in enumerate(lins):
                n_pad = mplan['n_pad'][li]
                kr.append(ops.mask_k_ranges(lin.mask, tn, (n_pad + tn - 1) // tn, mplan['k_pad'][li], mplan['row_of_out'][li],
                                            mplan['col_of_in'][li]))
            lp = mplan['layer_kernel'] = dict(k_ranges=kr, ld_scratch=max(mplan['n_pad'][:-1]))
        return lp, mplan

    def _forward_layer_kernel(self, x):
        x, ldx = _lib.rows(x, 'x')
        if ldx % 2 or x.data_ptr() % 8:
            x, ldx = x.contiguous(), x.shape[1]
        B, D = x.shape
        dev = x.device
        made, tr = self._conditioner, self._transformer
        made.begin_call()
        lp, mplan = self._layer_plan(dev)
        lins = made._linears()
        xs, xs_inv = ops.split_rows(x, mplan['k_pad'][0])
        y = torch.empty(B, D, dtype=torch.float32, device=dev)
        ldj = torch.empty(B, dtype=torch.float32, device=dev)
        scratch = [torch.empty(B, lp['ld_scratch'], dtype=torch.float32, device=dev) for _ in range(2 if len(lins) > 2 else 1)]
        d = _lib.MafLayerDesc()
        d.B, d.n_linears, d.kind = B, len(lins), 2
        d.a0, d.lda0, d.a0_inv_scale = xs.data_ptr(), xs.shape[1], xs_inv.data_ptr()
        keep = [xs, xs_inv]
        for li, lin in enumerate(lins):
            ws, w_sc, b, bmax = made._pack_layer_split(mplan, li, lin)
            keep += [ws, w_sc, b, bmax]
            d.w[li], d.ldw[li], d.n_rows_w[li] = ws.data_ptr(), ws.shape[1], ws.shape[0]
            d.n_out[li] = mplan['n_pad'][li] if li + 1 < len(lins) else lin.out_features
            d.w_scales[li], d.bias[li], d.bias_absmax[li] = w_sc.data_ptr(), b.data_ptr(), bmax.data_ptr()
            d.k_ranges[li] = lp['k_ranges'][li].data_ptr()
        d.scratch[0], d.scratch[1], d.ld_scratch = scratch[0].data_ptr(), scratch[-1].data_ptr(), lp['ld_scratch']
        d.x, d.ldx, d.y, d.ldy, d.log_det_J = x.data_ptr(), ldx, y.data_ptr(), D, ldj.data_ptr()
        d.n_features, d.moebius_dim, d.moebius_unit_sphere = D, 2, int(tr.unit_sphere)
        d.moebius_max_radius = tr.max_radius
        _lib.call('tfep_maf_layer_forward_split', ctypes.byref(d), _lib.stream_of(x))
        made.drop_packed_ahead()
        return y, ldj

    # ------------------------------------------------------------------ reference API
    def forward(self, x: torch.Tensor):
        """``(y, log_det_J)`` of the push-forward (reference autoregressive.py:144-177).

        Under autograd (grad mode on and the input or a conditioner parameter requires grad) the
        call is recorded as ONE graph node whose backward runs on the HIP kernels
        (``flows/_backward.py``); otherwise it is the plain forward.
        """
        ops.check_device_tensor(x, 'x')
        self._check_features(x, 'x')
        self._sync_conditioner()
        with self._range_guard(x):
            if torch.is_grad_enabled():
                from . import _backward
                params = _backward.trainable_tensors(self) if isinstance(self._conditioner, MADE) else \
                    [p for p in self._conditioner.parameters()]
                if x.requires_grad or any(p.requires_grad for p in params):
                    if _backward.supported(self):
                        return _backward.MAFLayerFunction.apply(self, x, *params)
                    if _backward.generic_supported(self):
                        return _backward.generic_forward(self, x)          # conditioner by autograd, transformer VJP kernel
                    return _backward.UnsupportedBackward.apply(self, x, *params)
            return self._forward_impl(x)

    def _check_features(self, x, name):
        """The kernels index ``x`` by the layer's own feature tables: a tensor of another width must never reach them
        (the reference fails in its first ``F.linear`` with a shape RuntimeError)."""
        n = self._inverse_masks.shape[1]
        if x.dim() < 1 or x.shape[-1] != n:
            raise RuntimeError(f'{type(self).__name__}: {name} has {x.shape[-1] if x.dim() else 0} features, the layer was '
                               f'built for {n} (degrees_in / dimension_in are those of the flow input, before any embedding)')

    #: Workgroups of the fused launches below which ``fused = None`` takes the generic path (exact-fp32 kernels only).
    fused_min_workgroups = 512

    def _fused_pays(self, x, kind):
        """Is the fused kernel the faster forward at this batch size?  Its point is that the (B, P D) parameter tensor never
        reaches HBM; its price, on the exact-fp32 kernel, is an epilogue in which every lane evaluates 8 features one after
        the other (2 on the affine tile) with nothing to overlap them.  With fewer workgroups than CUs that chain is the whole
        run time -- a 5-bin spline launch takes ~235 us however small -- while the generic path spreads the same
        evaluations over one wave per sample row (a 6-layer MixedMAFMap-like flow, D = 200, B = 1024: 9.1 ms fused, 4.0 ms
        generic).  Crossover, tools/probe/fused_crossover.py (one 8-bin layer, fused / generic ms): 256 workgroups 0.41 / 0.32
        and 1.59 / 1.26, 504 workgroups 3.71 / 3.66, 1024 workgroups 2.46 / 2.97, 4096 workgroups 5.7 / 7.6.  The
        split-f16 path is only taken for large products and always fuses."""
        if self.fused is not None:
            return bool(self.fused)
        B = x.shape[0]
        if self._use_split_gemm(B):
            return True
        fp = self._fused_plan(x.device, kind, self._tables(x.device))
        tm = 128                                             # rows per workgroup of gemm_kernel<2, ...>
        wgs = ((B + tm - 1) // tm) * sum(g['n_rows'] // (16 * g['P'] * g['FT']) for g in fp['groups'])
        return wgs >= int(os.environ.get('TFEP_FUSED_MIN_WGS', self.fused_min_workgroups))

    def _forward_impl(self, x: torch.Tensor):
        if self._layer_kernel_ok(x):
            return self._forward_layer_kernel(x)
        kind = self._fused_kind()
        if kind is not None and self._fused_pays(x, kind):
            return self._forward_fused(x, kind)
        parameters = self.get_transformer_parameters(x)
        tr = self._transformer
        if type(tr) is MoebiusTransformer and tr.dimension == 2 and tr.unit_sphere and not self.has_fixed_indices and \
                x.shape[1] % 8 == 0 and self._use_split_gemm(x.shape[0]) and os.environ.get('TFEP_SPLIT_HANDOVER', '1') != '0':
            # Outputs on the unit circle: the map also writes y as the split-f16 rows the NEXT layer's first GEMM reads (scale
            # of the bound |y| <= 1), and hands them over on the tensor -- no conversion pass between two such layers
            # (BASELINE cfg4-ii: 0.21 of 2.5 ms per layer).  ``MADE.forward`` takes them if the tensor is still the one.
            y, log_det_J, ys, ys_inv = ops.moebius_split_out(x, parameters, tr.max_radius, ops.round_up(x.shape[1], ops.tile_sizes()[2]))
            y._tfep_split = (ys, ys_inv, y._version)
            return y, log_det_J
        if self.has_fixed_indices:
            t = self._tables(x.device)
            y = x.clone()                               # fixed features propagate unchanged
            y_tr, log_det_J = self._transformer(ops.gather_columns(x, t['tr']), parameters)
            ops.scatter_columns(y_tr, t['tr'], y)
        else:
            y, log_det_J = self._transformer(x, parameters)
        return y, log_det_J

    def inverse(self, y: torch.Tensor):
        """``(x, log_det_J)`` of the inverse map (reference autoregressive.py:179-229).

        Under autograd (grad mode on and ``y`` or a parameter requires grad) the outputs are differentiable like the
        reference's (``_backward.LazyInverseFunction``: the values still come from the fast path, a backward pays for the
        reference's pass per degree).  With a MADE conditioner the values come from a blocked forward substitution: pass ``k``
        evaluates only the rows of the three masked linears that belong to degree ``k`` (contiguous
        row slices of the degree-sorted packed weights), so the whole inverse costs about one forward
        in flops instead of ``n_degrees`` forwards (a mixed transformer of affine / spline members: one
        step per degree and member).  Otherwise (user conditioner, another embedding,
        ``blocked_inverse=False``): one full conditioner pass per degree, like the reference; the
        last pass' log-det is the total.
        """
        ops.check_device_tensor(y, 'y')
        self._check_features(y, 'y')
        self._sync_conditioner()
        if torch.is_grad_enabled():
            params = [p for p in self.parameters() if p.requires_grad]
            if y.requires_grad or params:
                # Values from the fast path; a backward() that actually arrives re-runs the reference's own algorithm
                # (one conditioner pass per degree) built from differentiable pieces and back-propagates through it
                # (flows/_backward.py: LazyInverseFunction, generic_inverse).
                from . import _backward
                return _backward.LazyInverseFunction.apply(self, y, *params)
        return self._inverse_impl(y)

    def _inverse_impl(self, y: torch.Tensor):
        """The inverse under the range guard of the split-f16 default (``split_guard``).  The conditioner inputs of the inverse are
        its own OUTPUT, so the check comes after the fact: the feature scales of the x just computed decide -- one pass over x and one
        host synchronisation per call, on a path that takes tens of milliseconds -- and where they span more than 2^19 the call is
        repeated on the exact-fp32 kernels (``last_split_guard`` says so), the arithmetic a guarded forward of that x uses."""
        x, log_det_J = self._inverse_values(y)
        on = self.split_guard if self.split_guard is not None else os.environ.get('TFEP_SPLIT_GUARD', '1') != '0'
        if on and self.split_gemm is None and not self._guard_exact and y.shape[0] > 0 and self._use_split_gemm(y.shape[0]):
            if torch.cuda.is_current_stream_capturing():
                _flag_for_capture(x)
                return x, log_det_J
            n_x = ops.range_flag([ops.column_absmax(x.detach())], bits=19)
            self.last_split_guard = dict(feature_scales_out_of_range=bool(n_x), exact=bool(n_x), inverse=True)
            if n_x:
                self._guard_exact = True
                try:
                    x, log_det_J = self._inverse_values(y)
                finally:
                    self._guard_exact = False
                if not self.__dict__.get('_guard_warned'):
                    self.__dict__['_guard_warned'] = True
                    import warnings
                    warnings.warn('tfep_amd: the features of the inverse differ in scale by more than 2^19 (largest magnitude per '
                                  'feature over the batch): more than the split-f16 GEMMs carry at fp32 accuracy with one scale per '
                                  'row; this inverse call was repeated on the exact-fp32 MFMA kernels.  layer.split_gemm = True / '
                                  'False pins the arithmetic.')
        return x, log_det_J

    #: Rows per shard of a large blocked inverse (None: 8192; 0 / False: never shard).  The one-launch-per-super-block schedule needs
    #: every pair of chain / loader waves resident at once: 8192 rows at cfg2's sizes.  A larger batch used to fall back to the
    #: block-by-block schedule with one row per lane; it now runs shard after shard on the super-block kernel, the packed weights shared
    #: by the shards of the call (cfg2 layer, 65 536 rows: 533 -> 482 ms).  Above 16 384 rows only: up to there the whole batch runs on
    #: 16-row waves, which cfg4-i's short chains do faster than two shards (70 against 81 ms).
    inverse_shard_rows = None

    def _inverse_values(self, y: torch.Tensor):
        if self._blocked_ok():
            shard = self.inverse_shard_rows
            if os.environ.get('TFEP_INV_SHARD_ROWS') is not None:
                shard = int(os.environ['TFEP_INV_SHARD_ROWS'])
            shard = 8192 if shard is None else int(shard or 0)
            made = self._conditioner
            # (long chains only -- at least 1024 transformed features: where the chain of degrees dominates the call)
            if (shard > 0 and y.shape[0] > max(16384, shard) and isinstance(made, MADE) and self.__dict__.get('_shard_ok', True)
                    and y.shape[1] - len(self._conditioner_indices) >= 1024 and not torch.cuda.is_current_stream_capturing()):
                return self._inverse_blocked_sharded(y, shard)
            return self._inverse_blocked(y)
        t = self._tables(y.device)
        x = ops.zeros(*y.shape, dtype=y.dtype, device=y.device)
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

    # ------------------------------------------------------------------ blocked inverse
    def _blocked_ok(self):
        made = self._conditioner
        if not self.blocked_inverse or not isinstance(made, MADE) or len(self._conditioner_indices) > 0:
            return False
        if not made._degrees_ok:            # masks that no degree assignment reproduces: the reference's pass per degree
            return False
        emb = getattr(made, 'embedding', None)
        if (emb is not None and type(emb) is not PeriodicEmbedding) or len(made._linears()) < 2:
            return False
        tr = self._transformer
        if type(tr) in (AffineTransformer, NeuralSplineTransformer):
            return True
        if type(tr) is MoebiusTransformer:
            # every vector must live inside one degree (e.g. generate_degrees(..., repeats=dimension))
            deg = made._degrees[0]
            deg = deg[deg != -1]
            d = tr.dimension
            return len(deg) % d == 0 and bool((deg.reshape(-1, d) == deg.reshape(-1, d)[:, :1]).all())
        if type(tr) is MixedTransformer:
            # element-wise members: every degree becomes one step per member that has features in it
            return all(type(t) in (AffineTransformer, NeuralSplineTransformer) or self._is_plain_shift(t)
                       for t in tr._transformers)
        return False

    def _sub_transformer(self, sel, device, tr=None):
        """The transformer (``tr``: a member of a mixed transformer) restricted to its features ``sel`` (a degree group)."""
        tr = self._transformer if tr is None else tr
        if type(tr) is NeuralSplineTransformer:
            cfg, h = tr.config(device), tr.host()
            return ('spline', ops.SplineConfig(
                cfg.x0[sel], cfg.xf[sel], cfg.y0[sel], cfg.yf[sel], h['n_bins'], h['circular'], h['identity'],
                h['learn_lower'], h['learn_upper'], h['min_bin'], h['min_slope']))
        if type(tr) is MoebiusTransformer:
            return ('moebius', tr)
        if type(tr) is VolumePreservingShiftTransformer:
            return ('shift', tr)
        return ('affine', tr)

    #: Degrees per block of the two-level blocked inverse.
    inverse_block = 16

    #: Blocks per SUPER-BLOCK (third level, fused block kernel on split operands only): the contribution of every hidden
    #: unit older than the super-block to ALL rows of the super-block is one large GEMM per layer at its start; a block then
    #: adds only what the super-block itself has produced (a short GEMM), so the old activation panels are read once per
    #: super-block instead of once per block.  0 / 1: off.
    inverse_super = 8

    #: Run every block of a super-block in ONE launch (``tfep_inverse_block`` with ``n_blocks``): the pair of waves that owns
    #: 16 sample rows forms, at the head of each block, what the earlier blocks of the super-block add to the block's rows
    #: (exact-fp32 MFMA products on its own rows) instead of three short GEMM launches + two re-scaling launches per block in
    #: series with the chain.  None: whenever super-blocks and the paired kernel are in use (``TFEP_INV_SUPER_KERNEL=0``: off).
    inverse_super_kernel = None
    #: which schedule the last ``_inverse_blocked`` call ran: 'super_kernel' or 'block_by_block' (tests, probes)
    last_inverse_schedule = None

    def _input_columns(self):
        """Where feature column c of x enters the conditioner input: ``(first_col[c], periodic[c], limits)``.  Without an
        embedding the input IS x; a PeriodicEmbedding puts the non-periodic features first and then a (cos, sin) pair
        per periodic feature (mafembed.py:137-145)."""
        cached = self._dev.get('input_columns')         # (the index buffers live on the device: read them once -- a
        if cached is not None:                          # device -> host copy per call also cannot be captured in a graph)
            return cached
        emb = getattr(self._conditioner, 'embedding', None)
        D = self._inverse_masks.shape[1]
        if emb is None:
            res = list(range(D)), [False] * D, (0.0, 1.0)
        else:
            non, per = emb._nonperiodic_indices.tolist(), emb._periodic_indices.tolist()
            first, periodic = [0] * D, [False] * D
            for pos, c in enumerate(non):
                first[c] = pos
            for pos, c in enumerate(per):
                first[c], periodic[c] = len(non) + 2 * pos, True
            res = first, periodic, emb.host_limits()
        self._dev['input_columns'] = res
        return res

    def _input_info(self, x_cols, device):
        """Index tensors to write the features ``x_cols`` (a list, in the order of the value columns) into the
        conditioner-input buffer: plain features are copied, periodic ones become cos / sin."""
        first, periodic, _ = self._input_columns()
        i32 = dict(device=device, dtype=torch.int32)
        plain = [j for j, c in enumerate(x_cols) if not periodic[c]]
        per = [j for j, c in enumerate(x_cols) if periodic[c]]
        return dict(plain_sel=torch.tensor(plain, **i32), plain_cols=torch.tensor([first[x_cols[j]] for j in plain], **i32),
                    per_sel=torch.tensor(per, **i32), cos_cols=torch.tensor([first[x_cols[j]] for j in per], **i32),
                    sin_cols=torch.tensor([first[x_cols[j]] + 1 for j in per], **i32))

    def _scatter_inputs(self, xpad, values, info):
        """Write ``values`` (B, n) into the conditioner-input buffer as ``info`` (``_input_info``) says."""
        if info['plain_sel'].numel():
            ops.scatter_columns(ops.gather_columns(values, info['plain_sel']), info['plain_cols'], xpad)
        if info['per_sel'].numel():
            lo, hi = self._input_columns()[2]
            t = (ops.gather_columns(values, info['per_sel']) - lo) * float(2.0 * math.pi / (hi - lo))
            ops.scatter_columns(torch.cos(t), info['cos_cols'], xpad)
            ops.scatter_columns(torch.sin(t), info['sin_cols'], xpad)

    def _inverse_blocked_sharded(self, y, shard):
        """``_inverse_blocked`` shard by shard (rows are independent); the packs made for the first shard serve the others."""
        made = self._conditioner
        B = y.shape[0]
        x = torch.empty_like(y)
        ldj = torch.empty(B, dtype=torch.float32, device=y.device)
        keep = made.cache_packed_weights
        made.cache_packed_weights = True                   # (the packs outlive a shard's frozen_weights() block: checked by fingerprint)
        try:
            for r0 in range(0, B, shard):
                xs, ls = self._inverse_blocked(y[r0:r0 + shard])
                x[r0:r0 + shard] = xs
                ldj[r0:r0 + shard] = ls
                if self.last_inverse_schedule != 'super_kernel':
                    self.__dict__['_shard_ok'] = False      # this layer does not take the super-block kernel: whole batches from now on
        finally:
            made.cache_packed_weights = keep
            if not keep:
                for plan in made._plans.values():           # (as frozen_weights() leaves them without the cache)
                    for k in [k for k in plan if isinstance(k, tuple) and k[0] in ('packed', 'packed_split')]:
                        del plan[k]
        self.last_inverse_schedule = 'sharded ' + str(self.last_inverse_schedule)
        return x, ldj

    def _blocked_plan(self, device, batch=None):
        """The plan of ``_blocked_plan_for`` with ``inverse_block`` degrees per block, or fewer (halved down to 2) when
        that is what lets the block's state fit the LDS of the fused block kernel.  ``batch``: the rows of the call -- when
        the 16-row layout of the block kernel is not resident at once with ``inverse_block`` degrees per block but is with
        half of them (its LDS stage shrinks with the block), the halved plan is taken (cfg4-i at B = 16 384: 74.1 -> 70.1 ms)."""
        n_super = int(os.environ.get('TFEP_INV_SUPER', self.inverse_super or 0))
        G0 = max(1, int(os.environ.get('TFEP_INV_BLOCK', self.inverse_block)))
        if batch is not None and G0 >= 4 and 'TFEP_INV_BLOCK' not in os.environ and os.environ.get('TFEP_INV_BLOCK_BY_BATCH', '1') != '0':
            full = self._blocked_plan(device)
            # Round 4: the PAIRED 16-row kernel (and with it the one-launch-per-super-block schedule) needs every pair of the
            # call resident at once; its LDS grows with the degrees per block.  Where the full block does not fit twice per
            # CU but three quarters or half of it do, take the smaller block -- the super-block keeps its span of degrees
            # (cfg4-i at B = 8192: 85 KB per pair -> one pair per CU -> block by block, 56.2 ms; 12 degrees per block: 44.2).
            if full['fused'] is not None and n_super > 1 and self.inverse_paired is not False and \
                    os.environ.get('TFEP_INV_PAIRED', '1') != '0':
                lib = _lib.load()
                need = (int(batch) + 15) // 16

                def pairs_resident(bp_):
                    f_ = bp_['fused']
                    if f_ is None:
                        return False
                    ldsp = int(lib.tfep_inverse_block_lds_bytes_paired(bp_['L'], f_['cache_len'], f_['max_feats']))
                    return 0 < ldsp <= 160 * 1024 and need <= 256 * min((160 * 1024) // ldsp, 4)
                if not pairs_resident(full):
                    for g in (3 * G0 // 4, G0 // 2):
                        if g < 2 or g == G0:
                            continue
                        ns = max(2, int(round(n_super * G0 / g)))
                        pkey = ('blocked', str(device), g, ns)
                        cand = self._dev.get(pkey)
                        if cand is None:
                            self._plan_n_super = ns
                            cand = self._dev[pkey] = self._blocked_plan_for(device, g)
                        if pairs_resident(cand):
                            return cand
            if full['fused'] is not None and batch <= 16384:
                lib = _lib.load()
                need = (int(batch) + 15) // 16

                def resident(bp_):
                    f_ = bp_['fused']
                    lds16 = lib.tfep_inverse_block_lds_bytes_rows(bp_['L'], f_['cache_len'], f_['max_feats'], 16)
                    return 0 < lds16 <= 160 * 1024 and need <= 256 * ((160 * 1024) // int(lds16))
                if not resident(full):
                    hkey = ('blocked', str(device), G0 // 2, n_super)
                    half = self._dev.get(hkey)
                    if half is None:
                        self._plan_n_super = n_super
                        half = self._dev[hkey] = self._blocked_plan_for(device, G0 // 2)
                    if half['fused'] is not None and resident(half):
                        return half
            return full
        key = ('blocked', str(device), G0, n_super)
        bp = self._dev.get(key)
        if bp is not None:
            return bp
        self._plan_n_super = n_super
        G = G0
        bp = self._blocked_plan_for(device, G)
        if bp['fused'] is None and self._fused_inverse_supported(bp['L']):
            g = G
            while g > 2:
                g //= 2
                cand = self._blocked_plan_for(device, g)
                if cand['fused'] is not None:
                    bp = cand
                    break
        self._dev[key] = bp
        return bp

    def _blocked_plan_for(self, device, G):
        """Host-side plan of the two-level blocked forward substitution.

        Degrees are processed in blocks of ``inverse_block``.  For a block, the contribution of every
        EARLIER degree to all of the block's rows is one GEMM per layer over the (long) range of old
        hidden units -- the activation panel is read once per block instead of once per degree.  Inside
        the block each degree adds the (short) range of units that the block itself has produced.
        Hidden units are stored sorted by degree, so all of these are contiguous row / column ranges.
        """
        lib = _lib.load()
        made = self._conditioner
        mplan = made.plan(device)
        lins = made._linears()
        L = len(lins) - 1
        tk = lib.tfep_masked_linear_tile_k()
        tables = self._tables(device)
        deg_in = made._degrees[0].cpu()
        tr_idx = tables['tr'].cpu().long()
        # degrees in x space (deg_in lives in the conditioner-input space, which an embedding widens): feature c is
        # transformed at step d of the inverse iff _inverse_masks[d, c]
        deg_x = torch.full((self._inverse_masks.shape[1],), -1, dtype=torch.long)
        for d_, m_ in enumerate(self._inverse_masks.cpu()):
            deg_x[m_] = d_
        deg_tr = deg_x[tr_idx]
        n_tr = len(tr_idx)
        P = lins[-1].out_features // n_tr
        max_deg = int(deg_tr.max())
        hid = [torch.sort(made._degrees[l + 1].cpu()).values for l in range(L)]

        def up(v):
            return (v + tk - 1) // tk * tk

        def r_lo(l, e):      # first packed row of layer l with degree >= e
            return int(torch.searchsorted(hid[l], e, right=False))

        def r_hi(l, e):      # one past the last packed row of layer l with degree <= e
            return int(torch.searchsorted(hid[l], e, right=True))

        # output rows grouped by degree ("inverse packing"): base[d] .. base[d+1].  A degree is one STEP of the
        # substitution -- or, under a mixed transformer (parameters grouped by member, mixed.py:64-68), one step per
        # member with features of that degree: `parts[d]` = [(features among the transformed ones, P, member, the
        # features' positions inside the member, first output row)]
        sels = [torch.nonzero(deg_tr == d).flatten() for d in range(max_deg + 1)]
        tr = self._transformer
        mixed = type(tr) is MixedTransformer
        if mixed:
            splits = tr.host_splits() + [lins[-1].out_features]
            member_of = torch.full((n_tr,), -1, dtype=torch.long)
            local_of = torch.zeros(n_tr, dtype=torch.long)
            for g, ind in enumerate(tr._indices):
                ind = ind.cpu().long()
                member_of[ind] = g
                local_of[ind] = torch.arange(len(ind))
            n_of = [len(ind) for ind in tr._indices]
            P_of = [(splits[g + 1] - splits[g]) // max(n_of[g], 1) for g in range(len(n_of))]
            P = max(P_of)
        base, parts, cur = [0], [], 0
        row_inv = torch.empty(lins[-1].out_features, dtype=torch.long, device='cpu')
        for d, sel in enumerate(sels):
            if mixed:
                here = [(sel[member_of[sel] == g], P_of[g], g) for g in range(len(n_of))]
                here = [(s_, P_, g, local_of[s_]) for s_, P_, g in here if len(s_) > 0] or [(sel, P, None, None)]
            else:
                here = [(sel, P, None, None)]
            parts.append([])
            for s_, P_, g, loc in here:
                n_d = len(s_)
                for p in range(P_):
                    rows = (p * n_tr + s_) if g is None else (splits[g] + p * n_of[g] + loc)
                    row_inv[rows] = cur + p * n_d + torch.arange(n_d, device='cpu')
                parts[-1].append((s_, P_, g, loc, cur))
                cur += P_ * n_d
            base.append(cur)

        kr = []               # all k-ranges, one entry per launch: every tile of a launch shares it

        def rng(kb, ke):
            kr.append((kb, max(kb, ke)))
            return len(kr) - 1

        i32 = dict(device=device, dtype=torch.int32)
        blocks = []
        kA_prev = None
        n_super = max(1, int(getattr(self, '_plan_n_super', 0) or 1))
        supers, kS, dS0 = [], None, 0
        for d0 in range(0, max_deg + 1, G):
            d1 = min(d0 + G, max_deg + 1)
            blk = dict(wide=[], steps=[])
            # split point between "old" and "block" inputs of layer l >= 1 / the output layer:
            # old units of layer l-1 have degree <= d0 - 2
            kA = [0] + [(r_hi(l - 1, d0 - 2) // tk) * tk for l in range(1, L + 1)]
            # look-ahead split of the old range (see _inverse_blocked): [0, kP) was already complete before the PREVIOUS
            # block ran ('kr_old': its GEMM can overlap that block's kernel), [kP, kA) is what the previous block added
            # ('kr_new': a short GEMM afterwards)
            kP = kA if kA_prev is None else kA_prev

            # super-block (see ``inverse_super``): [0, kS) is what was complete when the block's super-block began
            if len(blocks) % n_super == 0:
                kS, dS0 = kA, d0
                dS1 = min(d0 + G * n_super, max_deg + 1)
                sup = dict(wide=[], out_wide=dict(layer=L, row0=base[dS0], n_rows=base[dS1] - base[dS0], kr=rng(0, kS[L])))
                for l in range(1, L):
                    r0, r1 = r_lo(l, dS0 - 1), r_hi(l, dS1 - 2)
                    if r1 > r0:
                        sup['wide'].append(dict(layer=l, row0=r0, n_rows=r1 - r0, kr=rng(0, kS[l])))
                # layer 0 of the whole super-block (the super-block kernel, ``inverse_super_kernel``): every conditioner input
                # known before the super-block -- bounding column range; what the super-block itself produces is still zero
                r0, r1 = r_lo(0, dS0 - 1), r_hi(0, dS1 - 2)
                known = torch.nonzero(deg_in <= dS0 - 1).flatten()
                kb0, ke0 = ((int(known.min()) // tk) * tk, min(up(int(known.max()) + 1), mplan['k_pad'][0])) if len(known) else (0, 0)
                sup['wide0'] = dict(layer=0, row0=r0, n_rows=max(r1 - r0, 0), kr=rng(kb0, ke0))
                sup['blocks'] = []
                supers.append(sup)
            blk['sb'], blk['sb_first'] = len(supers) - 1, len(blocks) % n_super == 0

            def wide_desc(l, r0, r1):
                return dict(layer=l, row0=r0, n_rows=r1 - r0, kr=rng(0, kA[l]), kr_old=rng(0, kP[l]),
                            kr_new=rng(kP[l], kA[l]), has_new=kA[l] > kP[l],
                            kr_sbnew=rng(kS[l], kA[l]), has_sbnew=kA[l] > kS[l], sb_range=(kS[l], kA[l]))
            for l in range(1, L):      # hidden layers fed by hidden layers
                r0, r1 = r_lo(l, d0 - 1), r_hi(l, d1 - 2)
                if r1 > r0:
                    blk['wide'].append(wide_desc(l, r0, r1))
            blk['out_wide'] = wide_desc(L, base[d0], base[d1])
            kA_prev = kA
            for d in range(d0, d1):
                e = d - 1
                hidden = []
                for l in range(L):
                    r0, r1 = r_lo(l, e), r_hi(l, e)
                    if r1 <= r0:
                        continue
                    if l == 0:     # layer 0 reads x in feature order: bounding range of the known columns
                        pos = torch.nonzero(deg_in <= e).flatten()
                        kb, ke = ((int(pos.min()) // tk) * tk, up(int(pos.max()) + 1)) if len(pos) else (0, 0)
                        ke = min(ke, mplan['k_pad'][0])
                    else:
                        kb, ke = kA[l], min(up(r_hi(l - 1, e)), mplan['k_pad'][l])
                    hidden.append(dict(layer=l, row0=r0, n_rows=r1 - r0, kr=rng(kb, ke)))
                ke = min(up(r_hi(L - 1, e)), mplan['k_pad'][L])
                # the device-side tables of a step (index tensors, sliced transformer) are only needed by the per-step
                # launches: built on first use (_step_tables), not for every degree of a layer the block kernel handles
                for i_, (sel, P_, g, loc, row0) in enumerate(parts[d]):
                    blk['steps'].append(dict(hidden=hidden if i_ == 0 else [],
                                             out=dict(row0=row0, n_rows=P_ * len(sel), kr=rng(kA[L], ke)),
                                             n_d=len(sel), sel_host=sel, cols_host=tr_idx[sel], member=g, local_host=loc))
            blk['fused'] = self._fused_block_tables(d0, d1, blk, kA, r_lo, r_hi, parts, tr_idx, deg_in, mplan, L,
                                                    rng, up, i32, d0_prev=d0 - G if d0 > 0 else None)
            blk['rows0'] = (r_lo(0, d0 - 1), max(r_hi(0, d1 - 2) - r_lo(0, d0 - 1), 0))      # layer-0 units of the block
            supers[-1]['blocks'].append(len(blocks))
            blocks.append(blk)
        narrow = lib.tfep_masked_linear_narrow_tile_n()
        max_rows = max([w['n_rows'] for b_ in blocks for w in b_['wide']] + [b_['out_wide']['n_rows'] for b_ in blocks] +
                       ([w['n_rows'] for s_ in supers for w in s_['wide'] + [s_['out_wide'], s_['wide0']]] if n_super > 1 else []) +
                       [b_['fused']['wide0']['n_rows'] for b_ in blocks if b_['fused'] and b_['fused']['wide0']] +
                       [h_['n_rows'] for b_ in blocks for st in b_['steps'] for h_ in st['hidden']] +
                       [st['out']['n_rows'] for b_ in blocks for st in b_['steps']])
        fused_ok = self._fused_inverse_supported(L) and all(b_['fused'] is not None for b_ in blocks)
        if fused_ok:
            cache_len = max(b_['fused']['cache_need'] for b_ in blocks)
            max_feats = max(b_['fused']['n_feats'] for b_ in blocks)
            fused_ok = 0 <= lib.tfep_inverse_block_lds_bytes(L, cache_len, max_feats) <= 160 * 1024
        bp = dict(blocks=blocks, P=P, L=L, row_inv=row_inv.to(**i32), n_rows_out=cur, supers=supers if n_super > 1 else None,
                  fused=dict(cache_len=cache_len, max_feats=max_feats) if fused_ok else None,
                  max_tiles=(max_rows + narrow - 1) // narrow,
                  k_ranges=torch.tensor(kr, dtype=torch.int32).reshape(-1, 2).to(device))
        return bp

    def _step_tables(self, st, device):
        """Device tensors of one degree for the per-step launches (cached in the step record)."""
        if 'sel' not in st:
            i32 = dict(device=device, dtype=torch.int32)
            sel, cols = st['sel_host'], st['cols_host']
            st['cols'] = cols.to(**i32)
            st['inputs'] = self._input_info(cols.tolist(), device)
            if st.get('member') is None:
                st['sub'] = self._sub_transformer(sel.to(device), device)
            else:       # a member of a mixed transformer, sliced by the features' positions inside the member
                st['sub'] = self._sub_transformer(st['local_host'].to(device), device,
                                                  tr=self._transformer._transformers[st['member']])
            st['sel'] = sel.to(**i32)
        return st

    #: Run the per-degree chain of each block in ONE kernel (``tfep_inverse_block``) when the layer qualifies.
    fused_inverse = True

    #: Sample rows per wave of the block kernel: 64 (one per lane) or 16 (four lanes per row); None: by batch size.
    inverse_rows_per_wave = None

    #: 16-row layout: a loader wave beside every chain wave (see ``_inverse_blocked``).  None: while every pair is resident.
    inverse_paired = None

    #: 16-row layout: independent waves per workgroup of the block kernel (1, 2, 4, 8; halved until the workgroup's LDS fits).
    #: More than one only packs the launch onto fewer CUs (see ``inverse_lookahead``).  None: one.
    inverse_waves_per_workgroup = None

    #: Overlap the wide GEMMs of the next block with the block kernel of the current one (side stream; results are the
    #: same sums in a different association: one more split-K slab).  None: when it pays (see ``_inverse_blocked``).
    inverse_lookahead = None

    #: The wide output-layer GEMM of every block (40 % of a cfg2 inverse) on split-f16 operands: None = when the forward
    #: uses them (``_use_split_gemm``) and a bound on |x| is known beforehand (see ``_split_inverse_bound``).
    split_inverse = None

    def _split_inverse_bound(self, device):
        """What is needed to bound |x| of the inverse before it is computed -- ``dict(dom, x0, xf, tail_slope)`` on the
        device -- or None when no bound is known in advance.

        The last hidden panel of the blocked inverse is filled block by block, so the row scale of its split-f16 copy
        must be fixed before the values exist: |h| is bounded layer by layer from a bound on the conditioner inputs.
        A spline with fixed bounds and the same domain and codomain maps y inside [x0, xf] to x inside it; outside, the
        map is linear with the boundary slope >= min_slope (spline.py:599-607: sentinel bins that continue the boundary
        slopes; slope 1 with identity boundary slopes), so |x| <= max(|x0|, |xf|) + (distance of y from the domain) /
        tail_slope.  Affine / Moebius outputs have no such bound (fp32 GEMMs)."""
        if self.split_inverse is False or not self.fused_inverse:
            return None
        if self.split_inverse is None and not self._use_split_gemm():
            return None
        key = ('split_inverse', str(device))
        if key not in self._dev:
            tr = self._transformer
            info = None
            if type(tr) is NeuralSplineTransformer:
                hst = tr.host()
                tail = 1.0 if (hst['identity'] or hst['circular']) else hst['min_slope']
                if not (hst['learn_lower'] or hst['learn_upper']) and tail > 1e-8 and \
                        bool(torch.equal(tr.x0, tr._y0)) and bool(torch.equal(tr.xf, tr._yf)):
                    f32 = dict(device=device, dtype=torch.float32)
                    info = dict(dom=torch.maximum(tr.x0.abs().max(), tr.xf.abs().max()).to(**f32),
                                x0=tr.x0.to(**f32).contiguous(), xf=tr.xf.to(**f32).contiguous(), tail_slope=float(tail))
            self._dev[key] = info
        return self._dev[key]

    def _fused_inverse_supported(self, L):
        if not self.fused_inverse or L > 4:
            return False
        tr = self._transformer
        if type(tr) is AffineTransformer:
            return True
        if type(tr) is MoebiusTransformer:
            return 1 <= tr.dimension <= 8
        if type(tr) is MixedTransformer:
            # kind 3 of the block kernel: spline members only (one instantiation per transformer family), each step of
            # the block names its member
            return len(tr._transformers) <= 8 and all(
                (type(t) is NeuralSplineTransformer and t.host()['n_bins'] <= 8) or self._is_plain_shift(t)
                for t in tr._transformers)
        return type(tr) is NeuralSplineTransformer and tr.host()['n_bins'] <= 8

    def _mixed_spline_descs(self, device):
        """``tfep_inverse_block`` kind 3: one spline descriptor per member of the mixed transformer, all of them over
        domain arrays laid out by TRANSFORMED FEATURE (the kernel indexes y and the domain with the same ``feat_sel``)."""
        key = ('mixed_descs', str(device))
        if key not in self._dev:
            tr = self._transformer
            n_tr = self._tables(device)['n_tr']
            arrays = [torch.zeros(n_tr, dtype=torch.float32, device=device) for _ in range(4)]
            arrays[1].fill_(1.0)
            arrays[3].fill_(1.0)
            for t, ind in zip(tr._transformers, tr._indices):
                if type(t) is NeuralSplineTransformer:
                    cfg = t.config(device)
                    ind = ind.to(device)
                    for dst, src in zip(arrays, (cfg.x0, cfg.xf, cfg.y0, cfg.yf)):
                        dst[ind] = src
            cfgs, descs = [], []
            for t in tr._transformers:
                if type(t) is NeuralSplineTransformer:
                    h = t.host()
                    cfgs.append(ops.SplineConfig(*arrays, h['n_bins'], h['circular'], h['identity'], h['learn_lower'],
                                                 h['learn_upper'], h['min_bin'], h['min_slope']))
                    descs.append(cfgs[-1].desc)
                else:       # the plain shift: n_bins = 0 (x = y - parameter, log-det 0)
                    descs.append(_lib.SplineDesc(*[a_.data_ptr() for a_ in arrays], 0, 0, 0, 0, 0, 0.0, 0.0))
            descs = (_lib.SplineDesc * len(descs))(*descs)
            self._dev[key] = (descs, cfgs, arrays)         # (keeps the arrays alive)
        return self._dev[key][0]

    def _fused_block_tables(self, d0, d1, blk, kA, r_lo, r_hi, parts, tr_idx, deg_in, mplan, L, rng, up, i32,
                            d0_prev=None):
        """Device tables of ``tfep_inverse_block`` for the block of degrees [d0, d1) (see include/tfep_hip.h)."""
        if not self._fused_inverse_supported(L):        # (e.g. a mixed transformer with an affine member: per-step launches)
            return None
        lib = _lib.load()
        tk = lib.tfep_masked_linear_tile_k()
        n_ints = lib.tfep_inverse_block_step_ints()
        # layer 0 gets a wide GEMM too: every feature of an earlier block (bounding column range; the block's own
        # features are still zero in the padded input when it runs)
        known = torch.nonzero(deg_in <= d0 - 1).flatten()
        if len(known):
            kb0, ke0 = (int(known.min()) // tk) * tk, min(up(int(known.max()) + 1), mplan['k_pad'][0])
        else:
            kb0, ke0 = 0, 0
        r0, r1 = r_lo(0, d0 - 1), r_hi(0, d1 - 2)
        wide0 = dict(layer=0, row0=r0, n_rows=r1 - r0, kr=rng(kb0, ke0), look=False) if r1 > r0 else None
        if wide0 is not None and d0_prev is not None:
            # look-ahead halves of the column range (see _inverse_blocked): 'kr_old' must not touch a column the PREVIOUS
            # block writes (its GEMM runs while that block does) and what is left must be one short range -- the case
            # for monotone degree orders; otherwise this block's layer-0 GEMM stays whole, after the previous block
            prev = torch.nonzero((deg_in >= d0_prev) & (deg_in <= d0 - 1)).flatten()
            if len(prev) and len(known) > len(prev):
                lo_p, hi_p = (int(prev.min()) // tk) * tk, min(up(int(prev.max()) + 1), ke0)
                cands = [(kb0, lo_p, lo_p, ke0), (hi_p, ke0, kb0, hi_p)]        # (old range, new range): prefix / suffix
                a0, a1, b0, b1 = max(cands, key=lambda c: c[1] - c[0])
                if a1 - a0 >= 4 * tk and b1 - b0 <= max(8 * tk, (ke0 - kb0) // 4):
                    wide0.update(look=True, kr_old=rng(a0, a1), kr_new=rng(b0, b1), has_new=b1 > b0)
        c0 = [kA[l + 1] for l in range(L)]
        n_old = [max(0, r_hi(l, d0 - 2) - c0[l]) for l in range(L)]
        cache_need = max([r_hi(l, d1 - 2) - c0[l] for l in range(L)] + [1])
        first, periodic, _ = self._input_columns()
        steps, cols, selv, feat_in, feat_per, in_cols = [], [], [], [], [], []
        for d in range(d0, d1):
          e = d - 1
          # one record per degree -- under a mixed transformer one per member with features of the degree, the hidden
          # units of the degree in the first of them
          for i_, (sel, _, member, _, row0) in enumerate(parts[d]):
            rec = [0] * n_ints
            for l in range(L):
                a, b = r_lo(l, e), r_hi(l, e)
                rec[4 * l], rec[4 * l + 1] = a, (max(0, b - a) if i_ == 0 else 0)
                if l == 0:
                    rec[2], rec[3] = 0, len(in_cols)          # conditioner-input entries of the block known so far
                else:
                    rec[4 * l + 2], rec[4 * l + 3] = c0[l - 1], max(c0[l - 1], r_hi(l - 1, e))
            if type(self._transformer) is MoebiusTransformer and len(sel) % self._transformer.dimension:
                return None                                   # a degree must hold whole vectors
            rec[16:22] = [row0, len(sel), c0[L - 1], max(c0[L - 1], r_hi(L - 1, e)), len(cols), member or 0]
            steps.append(rec)
            for c in tr_idx[sel].tolist():
                feat_in.append(len(in_cols))
                feat_per.append(int(periodic[c]))
                in_cols += [first[c], first[c] + 1] if periodic[c] else [first[c]]
            cols += tr_idx[sel].tolist()
            selv += sel.tolist()

        def dev_i32(v):
            return torch.tensor(v + [0], dtype=torch.int32).to(i32['device'])
        # units of every layer that THIS block computes: packed columns [lo, hi) of the activation panels
        unit_range = [(max(c0[l], r_hi(l, d0 - 2)), r_hi(l, d1 - 2)) for l in range(L)]
        return dict(wide0=wide0, c0=c0, n_old=n_old, cache_need=cache_need, n_feats=max(len(in_cols), 1), n_steps=len(steps),
                    unit_range=unit_range,
                    steps=torch.tensor(steps, dtype=torch.int32).reshape(-1, n_ints).to(i32['device']),
                    cols=dev_i32(cols), sel=dev_i32(selv), feat_in=dev_i32(feat_in), feat_per=dev_i32(feat_per),
                    in_cols=dev_i32(in_cols))

    #: The hidden-layer block GEMMs on split-f16 operands too (with ``split_inverse``): split copies of every hidden panel.
    split_inverse_hidden = True

    def _split_inverse_state(self, y, bp, mplan, lins, packs, h_last, n_out_max, y_tr=None, h_all=None):
        """Operands of the split-f16 output-layer block GEMM, or None when the layer does not qualify:
        ``(hs, hs_inv, w_split, w_inv, k_split)`` -- the (zeroed) split copy of the last hidden panel ``h_last`` with its
        bound-based per-row inverse scales, the split output weights in the inverse's row order, the number of slabs."""
        dev = y.device
        info = self._split_inverse_bound(dev)
        if info is None:
            return None
        made = self._conditioner
        L = bp['L']
        w_split, w_inv, _, _ = made._pack_layer_split(mplan, L, lins[L], row_of_out=bp['row_inv'], n_rows=bp['n_rows_out'])
        # |inputs| <= max(x bound, |fixed / conditioning features of y|, 1) (1: the cos / sin of a periodic embedding); per layer
        # |ELU(x W^T + b)| <= max(1, max|x| max_j sum_k |w_jk| + max|b|).  Reductions through the library: plain kernels,
        # nothing that becomes a memset node in a HIP graph (torch's multi-block reductions clear their semaphores with
        # hipMemsetAsync; see ops.zeros)
        y_tr = y if y_tr is None else y_tr
        outside = torch.clamp(torch.maximum(y_tr - info['xf'], info['x0'] - y_tr), min=0.0)     # distance from the domain
        x_bound = torch.maximum(ops.abs_reduce(y, 'row_max'), info['dom']) + \
            ops.abs_reduce(outside, 'row_max') * (1.0 / info['tail_slope'])
        bound = torch.clamp(x_bound, min=1.0)
        hidden = {}                 # layer l >= 1 fed by the hidden panel h[l - 1]: (split panel, its row scales, split W, W scale)
        # key 0 (round 4): the same for layer 0 and the padded conditioner input (the super-block kernel's schedule puts layer 0's
        # old inputs through one split GEMM per super-block): (row scales of the input panel, split W0, its scale); the split
        # copy of xpad is made by the caller
        if self.split_inverse_hidden and os.environ.get('TFEP_INV_SPLIT_LAYER0', '1') != '0':
            ws0, winv0 = made._pack_layer_split(mplan, 0, lins[0])[:2]
            hidden[0] = (None, ops.pow2_inv_scale(bound), ws0, winv0)
        for l in range(L):
            bound = torch.clamp(bound * ops.abs_reduce(packs[l][0], 'max_row_sum') +
                                ops.abs_reduce(packs[l][1].reshape(1, -1), 'row_max'), min=1.0)
            if l + 1 < L and self.split_inverse_hidden and h_all is not None:
                wsl, winvl = made._pack_layer_split(mplan, l + 1, lins[l + 1])[:2]
                hidden[l + 1] = (ops.zeros(*h_all[l].shape, dtype=torch.float32, device=dev), ops.pow2_inv_scale(bound), wsl, winvl)
        hs = ops.zeros(*h_last.shape, dtype=torch.float32, device=dev)          # filled block by block
        lib = _lib.load()
        tm, tn = lib.tfep_masked_linear_tile_m(), lib.tfep_masked_linear_tile_n()
        # split-K slabs: enough workgroups for 256 CUs (the split kernel's tile is 256 rows x 256 columns; the block kernel
        # fetches all slabs of a value in one round trip), >= 512 k per slice
        # column tile of the output block GEMM: 208 columns where the same number of tiles then carries less padding (the
        # 400 rows of 16 degrees x 25 parameters: 2 tiles, 4 % instead of 22 %)
        th = lib.tfep_split_half_wide_tile_n()
        n256, n208 = (n_out_max + 255) // 256, (n_out_max + th - 1) // th
        self._inv_out_tile = th if n208 <= n256 and os.environ.get('TFEP_INV_HALF_WIDE_TILE', '1') != '0' else 0
        positions = max(1, ((y.shape[0] + 255) // 256) * min(n256, n208))
        k_split = int(min(8, max(1, 256 // positions), max(1, mplan['k_pad'][L] // 512)))
        return hs, ops.pow2_inv_scale(bound), w_split, w_inv, k_split, hidden

    def _super_tables(self, bp, sup, L, device):
        """Concatenated step / feature tables and the per-block records of ``tfep_inverse_block`` (``n_blocks`` form) for one
        super-block, built once per plan."""
        st = sup.get('tables')
        if st is not None:
            return st
        lib = _lib.load()
        n_rec = lib.tfep_inverse_block_record_ints()
        blocks = [bp['blocks'][i] for i in sup['blocks']]
        steps, cols, sel, feat_in, feat_per, in_cols, recs = [], [], [], [], [], [], []
        n_steps = n_feat = n_in = 0
        in0 = 0                                           # first input entry of the super-block in the concatenated table
        for blk in blocks:
            fb = blk['fused']
            nf, ni = fb['cols'].numel() - 1, fb['in_cols'].numel() - 1       # (the per-block tables carry one padding entry)
            rec = [0] * n_rec
            rec[0], rec[1], rec[2], rec[3] = fb['n_steps'], n_steps, n_feat, n_in
            rec[32] = nf                                   # features of the block (the kernel's per-feature LDS table)
            for l in range(L):
                rec[4 + l], rec[8 + l] = fb['c0'][l], fb['n_old'][l]
            # products at the head of the block: layer 0 from the super-block's earlier input entries, layer l >= 1 from the
            # packed columns [kS, kA) of layer l - 1
            r0, n0 = blk['rows0']
            rec[12:16] = [r0, n0, in0, n_in]
            for wd in blk['wide'] + [blk['out_wide']]:
                l = wd['layer']
                rec[12 + 4 * l:16 + 4 * l] = [wd['row0'], wd['n_rows'], wd['sb_range'][0], wd['sb_range'][1]]
            recs.append(rec)
            steps.append(fb['steps'])
            cols.append(fb['cols'][:nf]); sel.append(fb['sel'][:nf]); feat_in.append(fb['feat_in'][:nf]); feat_per.append(fb['feat_per'][:nf])
            in_cols.append(fb['in_cols'][:ni])
            n_steps += fb['n_steps']; n_feat += nf; n_in += ni
        pad = torch.zeros(1, dtype=torch.int32, device=device)
        cat = lambda v: torch.cat(v + [pad]).contiguous()
        ic = torch.cat(in_cols).cpu() if in_cols else torch.zeros(0, dtype=torch.int32)
        per_any = bool(torch.cat(feat_per).any().item()) if feat_per else False
        in_range = (int(ic.min()), int(ic.max()) + 1) if len(ic) else (0, 0)        # (a periodic entry list names both columns)
        st = sup['tables'] = dict(
            in_range=in_range, periodic=per_any, max_steps=max(b_['fused']['n_steps'] for b_ in blocks),
            steps=torch.cat(steps).contiguous(), cols=cat(cols), sel=cat(sel), feat_in=cat(feat_in), feat_per=cat(feat_per),
            in_cols=cat(in_cols), records=torch.tensor(recs, dtype=torch.int32).reshape(-1, n_rec).to(device), n_blocks=len(blocks),
            unit_range=[(blocks[0]['fused']['unit_range'][l][0], blocks[-1]['fused']['unit_range'][l][1]) for l in range(L)])
        return st

    def _inverse_blocked(self, y):
        y, _ = _lib.rows(y, 'y')
        B, D = y.shape
        dev = y.device
        tables = self._tables(dev)
        bp = self._blocked_plan(dev, batch=B)
        made = self._conditioner
        mplan = made.plan(dev)
        lins = made._linears()
        L = bp['L']
        narrow = _lib.load().tfep_masked_linear_narrow_tile_n()
        f32 = dict(dtype=torch.float32, device=dev)
        kr_all = bp['k_ranges']

        def launch(x_in, w, bias, desc, out, out_col0, act, pre=None, pre_col0=0, wide=False, k_split=1, split=None, tile_n=0):
            """out[:, out_col0 : +n] = act(x_in W[row0 : row0+n]^T + bias[row0:] (+ pre[:, pre_col0 : +n]));
            ``k_split`` > 1: ``out`` is (k_split, B, cols) and receives the partial sums of the k slices;
            ``split = (x_inv_scale, w_inv_scale)``: ``x_in`` and ``w`` are split-f16 rows (wide tile only)."""
            n, row0 = desc['n_rows'], desc['row0']
            d = _lib.GemmDesc()
            if split is not None:
                d.split, d.x_inv_scale, d.w_inv_scale = 1, split[0].data_ptr(), split[1].data_ptr()
                wide = True
            d.x, d.ldx = x_in.data_ptr(), x_in.shape[1]
            d.w, d.ldw = w.data_ptr() + 4 * row0 * w.shape[1], w.shape[1]
            d.bias = (bias.data_ptr() + 4 * row0) if bias is not None else None
            d.k_ranges = krs[desc['kr']].data_ptr()
            d.y, d.ldy = out.data_ptr() + 4 * out_col0, out.shape[-1]
            d.B, d.N, d.n_rows_w, d.k_padded, d.act, d.accumulate = B, n, n, w.shape[1], act, 0
            if k_split > 1:
                d.k_split, d.slab_stride = k_split, out.shape[-2] * out.shape[-1]
            if pre is not None:
                d.pre_add, d.ld_pre_add = pre.data_ptr() + 4 * pre_col0, pre.shape[1]
            d.tile_n = (tile_n if split is not None else 0) if wide else narrow
            _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), _lib.stream_of(x_in))

        # every column tile of a launch shares the launch's k-range: one row-repeated table, built once
        krs = bp.get('kr_tiles')
        if krs is None:
            krs = kr_all[:, None, :].expand(-1, bp['max_tiles'], 2).contiguous()
            bp['kr_tiles'] = krs

        with made.frozen_weights():
            if bp['fused'] is not None and self._split_inverse_bound(dev) is not None:
                # the block kernel reads fp32 weights, the block GEMMs split ones: both from one pass over the parameters
                for l in range(1, L):
                    if self.split_inverse_hidden:
                        made._pack_layer_both(mplan, l, lins[l])
                made._pack_layer_both(mplan, L, lins[L], row_of_out=bp['row_inv'], n_rows=bp['n_rows_out'])
            packs = [made._pack_layer(mplan, l, lins[l]) for l in range(L)]
            w_out, b_out = made._pack_layer(mplan, L, lins[L], row_of_out=bp['row_inv'], n_rows=bp['n_rows_out'])
            x = ops.zeros(B, D, **f32)
            xpad = ops.zeros(B, mplan['k_pad'][0], **f32)          # conditioner input, zero padded
            if self.has_fixed_indices:
                fixed = ops.gather_columns(y, tables['fixed'])
                ops.scatter_columns(fixed, tables['fixed'], x)
                if 'fixed_inputs' not in bp:
                    bp['fixed_inputs'] = self._input_info(tables['fixed'].tolist(), dev)
                self._scatter_inputs(xpad, fixed, bp['fixed_inputs'])
                y_tr = ops.gather_columns(y, tables['tr'])
            else:
                y_tr = y
            h = [ops.zeros(B, mplan['n_pad'][l], **f32) for l in range(L)]
            z = [None] + [torch.empty(B, mplan['n_pad'][l], **f32) for l in range(1, L)]   # partial pre-activations
            zout = torch.empty(B, bp['n_rows_out'], **f32)
            ldj = ops.zeros(B, **f32)
            fused = bp['fused']
            if fused is not None:
                # The block GEMMs are short and wide (B x ~100 rows of W over up to 15 000 k): too few output tiles for
                # 256 CUs, so they run split-K into S slabs that the block kernel adds up.
                tm = ops.tile_sizes()[0]
                m_tiles = (B + tm - 1) // tm
                S = int(min(8, max(1, 256 // max(1, 2 * m_tiles)), max(1, max(mplan['k_pad']) // 512)))   # >= 512 k per slice
                if self.split_inverse_hidden and self._split_inverse_bound(dev) is not None and L > 1:
                    S = int(min(8, max(1, 256 // max(1, (B + 255) // 256)), max(1, max(mplan['k_pad']) // 512)))   # 256-row tiles
                if os.environ.get('TFEP_INV_SLABS'):
                    S = int(os.environ['TFEP_INV_SLABS'])
                # slabs hold only the block's own rows: column c of a slab is packed row (first row of the block + c)
                wz = [1] * L
                wzout = 1
                for b_ in bp['blocks']:
                    for wd in b_['wide'] + ([b_['fused']['wide0']] if b_['fused']['wide0'] is not None else []):
                        wz[wd['layer']] = max(wz[wd['layer']], wd['n_rows'])
                    wzout = max(wzout, b_['out_wide']['n_rows'])
                # 16 sample rows per wave (4x the waves) while the batch leaves SIMDs idle: B / 64 one-row-per-lane
                # waves fill at most a quarter of the 1024 SIMDs up to 16 384 rows (see csrc/inverse_block.hip)
                rows = self.inverse_rows_per_wave
                if os.environ.get('TFEP_INV_ROWS_PER_WAVE'):
                    rows = int(os.environ['TFEP_INV_ROWS_PER_WAVE'])
                if not rows:
                    # ... and only while every 16-row workgroup is resident at once (LDS per workgroup: the weight stage
                    # does not shrink with the rows): cfg4-i at B = 16 384 would need 4 per CU at 85 KB each
                    lds16 = _lib.load().tfep_inverse_block_lds_bytes_rows(L, fused['cache_len'], fused['max_feats'], 16)
                    fit = (160 * 1024) // max(int(lds16), 1)
                    rows = 16 if B <= 16384 and 0 < lds16 and (B + 15) // 16 <= 256 * fit else 64
                rows_per_wave = int(rows)
                # look-ahead: the long "old" part of block k + 1's wide GEMMs runs on a side stream WHILE block k's kernel
                # (one wave per 64 samples: half the CUs at batch 8192) runs; what block k added follows as one short
                # GEMM into an extra slab.  Two sets of slabs, alternating between blocks.
                pack4 = False
                # round 3: a LOADER wave beside every 16-row wave (``tfep_inverse_block`` paired): staging the weights and the
                # block GEMMs' slabs was 48 % of the chain's instructions, and a lone wave issues one vector instruction per
                # ~8 cycles -- the second wave takes that half off the chain.  While every pair is resident at once.
                paired = self.inverse_paired
                if os.environ.get('TFEP_INV_PAIRED') is not None:
                    paired = os.environ['TFEP_INV_PAIRED'] != '0'
                if rows_per_wave != 16:
                    paired = False
                elif paired is None:
                    ldsp = _lib.load().tfep_inverse_block_lds_bytes_paired(L, fused['cache_len'], fused['max_feats'])
                    fitp = (160 * 1024) // max(int(ldsp), 1)
                    paired = 0 < ldsp and fitp >= 1 and (B + 15) // 16 <= 256 * min(fitp, 4)
                look = self.inverse_lookahead
                if os.environ.get('TFEP_INV_LOOKAHEAD') is not None:
                    look = os.environ['TFEP_INV_LOOKAHEAD'] != '0'
                if look is None:
                    # pays when the block kernel leaves CUs free (one wave per 64 samples, one workgroup per CU) and the
                    # GEMMs are long enough to be worth two event round trips per block: cfg2 layer at B = 8192
                    # 150 -> 138 ms, neutral at B = 16 384 (all 256 CUs taken); cfg1 (launch bound) +7 %, cfg4-i at
                    # B = 16 384 +11 % with it -- those stay in order
                    # 16-row waves sit on every CU: the GEMMs (a whole SIMD's registers per wave) find no room beside them
                    # (cfg2 layer at B = 8192, hidden GEMMs on split operands: 125.5 ms in order, 140 ms with look-ahead)
                    look = rows_per_wave == 64 and (B + 63) // 64 <= 160 and max(mplan['k_pad']) >= 4096
                    # round 3: 16-row waves PACKED four to a workgroup (each wave its own LDS region, 39 KB at cfg2: four
                    # fill a CU's LDS) leave whole CUs to the GEMMs: cfg2 layer at B = 8192 107.3 -> 102.5 ms (the block
                    # kernel on 128 CUs, the look-ahead GEMMs on the other 128 now bound the block: 8 waves per CU would
                    # need half the LDS per wave; tools/probe/inv_pack.py, profiles/r03_inverse_pack.txt)
                    if not look and not paired and rows_per_wave == 16 and self.inverse_waves_per_workgroup is None and \
                            not os.environ.get('TFEP_INV_WPW') and max(mplan['k_pad']) >= 4096:
                        lds16 = _lib.load().tfep_inverse_block_lds_bytes_rows(L, fused['cache_len'], fused['max_feats'], 16)
                        n_waves = (B + 15) // 16
                        if 0 < 4 * lds16 <= 160 * 1024 and 128 < n_waves <= 4 * 160:
                            # onto ~128 CUs: two waves per workgroup up to 4096 rows (92.2 -> 86.1 ms), four up to 10 240
                            look, pack4 = True, (2 if n_waves <= 256 else 4)
                look = bool(look) and len(bp['blocks']) > 1
                n_par = 2 if look else 1
                zs = [[torch.empty(S + 1, B, ops.round_up(wz[l], 4), **f32) for l in range(L)] for _ in range(n_par)]
                z = zs[0]
                S_out = S
                tr = self._transformer
                kind = {NeuralSplineTransformer: 1, MoebiusTransformer: 2, MixedTransformer: 3}.get(type(tr), 0)
                spl = tr.config(dev).desc if kind == 1 else None
                d = _lib.InverseBlockDesc()
                d.B, d.n_layers, d.kind = B, L, kind
                if kind == 3:       # one descriptor per member; the step records name the member
                    spl = self._mixed_spline_descs(dev)
                    d.n_spline_groups = len(spl)
                if kind == 2:
                    d.moebius_dim, d.moebius_unit_sphere = tr.dimension, int(tr.unit_sphere)
                    d.moebius_max_radius = tr.max_radius
                d.x, d.ldx, d.xpad, d.ldxpad = x.data_ptr(), D, xpad.data_ptr(), xpad.shape[1]
                d.y, d.ldy = y_tr.data_ptr(), y_tr.shape[1]
                for l in range(L):
                    d.h[l], d.ldh[l] = h[l].data_ptr(), h[l].shape[1]
                    d.z[l], d.ldz[l] = z[l].data_ptr(), z[l].shape[-1]
                    d.z_slab_stride[l] = B * z[l].shape[-1]
                    d.w[l], d.ldw[l] = packs[l][0].data_ptr(), packs[l][0].shape[1]
                d.log_det_J = ldj.data_ptr()
                d.wout, d.ldwout = w_out.data_ptr(), w_out.shape[1]
                d.cache_len, d.max_feats = fused['cache_len'], fused['max_feats']
                d.rows_per_wave = rows_per_wave
                wpw = self.inverse_waves_per_workgroup
                if os.environ.get('TFEP_INV_WPW'):
                    wpw = int(os.environ['TFEP_INV_WPW'])
                if wpw is None and pack4 and look:
                    wpw = int(pack4)
                if paired:
                    wpw, d.paired = None, 1
                if rows_per_wave == 16 and wpw:
                    lds16 = _lib.load().tfep_inverse_block_lds_bytes_rows(L, fused['cache_len'], fused['max_feats'], 16)
                    while wpw > 1 and wpw * lds16 > 160 * 1024:
                        wpw //= 2
                    d.waves_per_workgroup = max(1, int(wpw))
                d.spline = None if spl is None else ctypes.cast(spl if kind == 3 else ctypes.pointer(spl), ctypes.c_void_p)
                d.emb_lower, d.emb_upper = self._input_columns()[2]
                stream = _lib.stream_of(y)
                # ---- the output-layer block GEMM on split-f16 operands (see _split_inverse_bound)
                hs, hs_hidden = None, {}
                sp = self._split_inverse_state(y, bp, mplan, lins, packs, h[L - 1], wzout, y_tr=y_tr, h_all=h)
                if sp is not None:
                    hs, hs_inv, ws_out, winv_out, S_out, hs_hidden = sp
                zouts = [torch.empty(S_out + 1, B, ops.round_up(wzout, 4), **f32) for _ in range(n_par)]
                d.ldzout, d.zout_slab_stride = zouts[0].shape[-1], B * zouts[0].shape[-1]

                def wide_gemms(blk, par, part):
                    """The wide GEMMs of a block over the old hidden units into the slabs of parity ``par``:
                    ``part`` 'kr' = the whole old range, 'kr_old' / 'kr_new' = its look-ahead halves."""
                    new_part = part == 'kr_new'
                    w0 = blk['fused']['wide0']
                    for wd in blk['wide'] + ([w0] if w0 is not None and (part == 'kr' or w0['look']) else []) + \
                            [blk['out_wide']]:
                        if new_part and not wd['has_new']:
                            continue
                        l = wd['layer']
                        desc = dict(wd, kr=wd[part])
                        if l < L and l in hs_hidden and not new_part:
                            hsl, hsl_inv, wsl, winvl = hs_hidden[l]
                            launch(hsl, wsl, packs[l][1], desc, zs[par][l], 0, act=0, k_split=S, split=(hsl_inv, winvl))
                        elif l < L:
                            out, ks = (zs[par][l][S], 1) if new_part else (zs[par][l], S)
                            launch(h[l - 1] if l > 0 else xpad, packs[l][0], None if new_part else packs[l][1], desc, out, 0,
                                   act=0, k_split=ks)
                        elif hs is not None and not new_part:
                            out, ks = (zouts[par][S_out], 1) if new_part else (zouts[par], S_out)
                            launch(hs, ws_out, None if new_part else b_out, desc, out, 0, act=0, k_split=ks,
                                   split=(hs_inv, winv_out), tile_n=getattr(self, '_inv_out_tile', 0))
                        else:
                            out, ks = (zouts[par][S_out], 1) if new_part else (zouts[par], S_out)
                            launch(h[L - 1], w_out, None if new_part else b_out, desc, out, 0, act=0,
                                   wide=wd['n_rows'] > 4 * narrow, k_split=ks)

                # ---- super-blocks (``inverse_super``): old panels read once per super-block.  Per layer one buffer of
                # S_sb + 1 slabs in the super-block's row layout: slabs [0, S_sb) = the large GEMM over [0, kS) for every row of
                # the super-block (split-K), slab S_sb = the block's short GEMM over [kS, kA), written at the block's columns.
                supers = bp.get('supers')
                x_split = hs_hidden.pop(0, None)           # (layer 0 on split operands: the super-block kernel's schedule only)
                use_sb = bool(supers) and hs is not None and not look and all(l in hs_hidden for l in range(1, L)) and \
                    all(wd['layer'] >= 1 for b_ in bp['blocks'] for wd in b_['wide'])
                if os.environ.get('TFEP_INV_DEBUG'):
                    print('inverse schedule:', dict(B=B, supers=bool(supers), split_state=hs is not None, look=look, paired=paired,
                                                    rows_per_wave=rows_per_wave, hidden_split=sorted(hs_hidden), use_sb=use_sb,
                                                    cache_len=fused['cache_len'], max_feats=fused['max_feats'], blocks=len(bp['blocks'])))
                sk_wanted = self.inverse_super_kernel
                if os.environ.get('TFEP_INV_SUPER_KERNEL') is not None:
                    sk_wanted = os.environ['TFEP_INV_SUPER_KERNEL'] != '0'
                sk_wanted = (sk_wanted is None or bool(sk_wanted)) and bool(paired) and rows_per_wave == 16
                if use_sb:
                    tw = ops.split_wide_tile_n()
                    m_tiles256 = (B + 255) // 256
                    sb_rows = {l: max([w['n_rows'] for s_ in supers for w in s_['wide'] if w['layer'] == l] or [0]) for l in range(1, L)}
                    sb_rows[L] = max(s_['out_wide']['n_rows'] for s_ in supers)
                    sb_tile = {l: (tw if (-sb_rows[l]) % tw <= (-sb_rows[l]) % 256 else 0) for l in sb_rows}
                    sb_S, sb_buf = {}, {}
                    for l, n_l in sb_rows.items():
                        if n_l == 0:
                            continue
                        n_tiles = (n_l + (sb_tile[l] or 256) - 1) // (sb_tile[l] or 256)
                        sb_S[l] = int(min(8, max(1, 256 // max(1, m_tiles256 * n_tiles)), max(1, mplan['k_pad'][l] // 512)))
                        if sk_wanted:
                            sb_S[l] = min(sb_S[l], 3)       # (+ the products' slab: one pass of the chain loader's stage)
                        sb_buf[l] = torch.empty(sb_S[l] + 1, B, ops.round_up(n_l, 4), **f32)

                    def super_gemms(sup):
                        for wd in sup['wide'] + [sup['out_wide']]:
                            l = wd['layer']
                            if wd['n_rows'] == 0:
                                continue
                            out = sb_buf[l][:sb_S[l]] if sb_S[l] > 1 else sb_buf[l][0]
                            if l < L:
                                hsl, hsl_inv, wsl, winvl = hs_hidden[l]
                                launch(hsl, wsl, packs[l][1], wd, out, 0, act=0, k_split=sb_S[l], split=(hsl_inv, winvl),
                                       tile_n=sb_tile[l])
                            else:
                                launch(hs, ws_out, b_out, wd, out, 0, act=0, k_split=sb_S[l], split=(hs_inv, winv_out),
                                       tile_n=sb_tile[l])

                    def block_gemms_sb(blk, sup):
                        """The short GEMMs of a block over what its super-block has produced so far, into the extra slab."""
                        for wd in blk['wide'] + [blk['out_wide']]:
                            if not wd['has_sbnew']:
                                continue
                            l = wd['layer']
                            sd = sup['out_wide'] if l == L else [w for w in sup['wide'] if w['layer'] == l][0]
                            desc = dict(wd, kr=wd['kr_sbnew'])
                            if l < L:
                                hsl, hsl_inv, wsl, winvl = hs_hidden[l]
                                launch(hsl, wsl, None, desc, sb_buf[l][sb_S[l]], wd['row0'] - sd['row0'], act=0,
                                       split=(hsl_inv, winvl))
                            else:
                                launch(hs, ws_out, None, desc, sb_buf[l][sb_S[l]], wd['row0'] - sd['row0'], act=0,
                                       split=(hs_inv, winv_out), tile_n=getattr(self, '_inv_out_tile', 0))
                if look:
                    main, side = torch.cuda.current_stream(dev), _side_stream(dev)
                    wide_gemms(bp['blocks'][0], 0, 'kr_old')
                    old_done = None
                # ---- ONE launch per super-block (``inverse_super_kernel``)
                sk = self.inverse_super_kernel
                if os.environ.get('TFEP_INV_SUPER_KERNEL') is not None:
                    sk = os.environ['TFEP_INV_SUPER_KERNEL'] != '0'
                steps_fit = all(b_['fused']['n_steps'] <= ops.round_up(fused['max_feats'], 4) for b_ in bp['blocks'])
                if (sk is None or sk) and use_sb and paired and rows_per_wave == 16 and steps_fit:
                    # layer 0 joins the super-block scheme: its own slabs, one GEMM per super-block over the inputs known before
                    n0_max = max(s_['wide0']['n_rows'] for s_ in supers)
                    # split-K slices of layer 0's GEMM: one round of workgroups, and at most 3 -- with the slab of the in-kernel products
                    # the chain's loader then adds 4 slabs, one pass of its stage (5 slabs: two dependent passes in every layer-0
                    # slot of the chain; cfg2 layer, B = 8192: 4 -> 2 slices, 68.8 / 69.1 -> 67.4 / 67.8 ms)
                    S0 = int(max(1, min(3, 256 // max(1, m_tiles * ((n0_max + 255) // 256)), mplan['k_pad'][0] // 512)))
                    if os.environ.get('TFEP_INV_S0'):
                        S0 = int(os.environ['TFEP_INV_S0'])
                    sb_buf0 = torch.empty(S0 + 1, B, ops.round_up(max(n0_max, 1), 4), **f32)
                    xs = None
                    if x_split is not None:
                        # split copy of the padded conditioner input, with the row scale of the bound on |x|: what is there
                        # already (fixed / conditioning features), then the columns every super-block adds
                        _, xs_inv, ws0, winv0 = x_split
                        xs = ops.zeros(*xpad.shape, **f32)
                        ops.split_columns_scaled(xpad, 0, xpad.shape[1] // 8 * 8, xs, xs_inv)
                    # two pairs per workgroup (they share the weight fetches of the in-kernel products) while every workgroup is
                    # still resident: two pairs' LDS on one CU
                    ldsp = int(_lib.load().tfep_inverse_block_lds_bytes_paired(L, fused['cache_len'], fused['max_feats']))
                    fit2 = (160 * 1024) // max(2 * ldsp, 1)
                    two = fit2 >= 1 and ((B + 15) // 16 + 1) // 2 <= 256 * min(fit2, 2) and os.environ.get('TFEP_INV_TWO_PAIRS', '1') != '0'
                    d.rows_per_wave, d.paired, d.waves_per_workgroup = 16, 1, (4 if two else 0)
                    self.last_inverse_schedule = 'super_kernel'
                    if os.environ.get('TFEP_INV_SPLIT_PRODUCTS', '1') != '0':
                        # the in-kernel products on split-f16 operands, like the short GEMMs they replace: the layers' split
                        # packs and the row scales of the panels' split copies
                        for l_, (hsl, hsl_inv, wsl, winvl) in hs_hidden.items():
                            d.ws[l_], d.ldws[l_], d.ws_inv_scale[l_], d.h_inv_scale[l_] = wsl.data_ptr(), wsl.shape[1], winvl.data_ptr(), hsl_inv.data_ptr()
                        d.ws[L], d.ldws[L], d.ws_inv_scale[L], d.h_inv_scale[L] = ws_out.data_ptr(), ws_out.shape[1], winv_out.data_ptr(), hs_inv.data_ptr()
                    for sup in supers:
                        tb = self._super_tables(bp, sup, L, dev)
                        super_gemms(sup)
                        w0 = sup['wide0']
                        if w0['n_rows'] > 0 and xs is not None:
                            launch(xs, ws0, packs[0][1], w0, sb_buf0[:S0] if S0 > 1 else sb_buf0[0], 0, act=0, k_split=S0, split=(xs_inv, winv0))
                        elif w0['n_rows'] > 0:
                            launch(xpad, packs[0][0], packs[0][1], w0, sb_buf0[:S0] if S0 > 1 else sb_buf0[0], 0, act=0, k_split=S0)
                        d.n_blocks, d.blocks = tb['n_blocks'], tb['records'].data_ptr()
                        d.n_steps = tb['max_steps']                 # (the kernel keeps a block's step records in LDS)
                        d.steps, d.feat_cols, d.feat_sel = tb['steps'].data_ptr(), tb['cols'].data_ptr(), tb['sel'].data_ptr()
                        d.feat_in, d.feat_periodic, d.in_cols = tb['feat_in'].data_ptr(), tb['feat_per'].data_ptr(), tb['in_cols'].data_ptr()
                        # the kernel indexes the slabs by packed row: column 0 of a buffer is the super-block's first row
                        d.z[0] = sb_buf0.data_ptr() - 4 * w0['row0']
                        d.ldz[0], d.z_slab_stride[0], d.z_slabs[0] = sb_buf0.shape[-1], B * sb_buf0.shape[-1], S0
                        d.z_extra[0] = d.z[0] + 4 * S0 * d.z_slab_stride[0]
                        for sd in sup['wide']:
                            l = sd['layer']
                            d.z[l] = sb_buf[l].data_ptr() - 4 * sd['row0']
                            d.ldz[l], d.z_slab_stride[l], d.z_slabs[l] = sb_buf[l].shape[-1], B * sb_buf[l].shape[-1], sb_S[l]
                            d.z_extra[l] = d.z[l] + 4 * sb_S[l] * d.z_slab_stride[l]
                        d.zout = sb_buf[L].data_ptr() - 4 * sup['out_wide']['row0']
                        d.ldzout, d.zout_slab_stride, d.zout_slabs = sb_buf[L].shape[-1], B * sb_buf[L].shape[-1], sb_S[L]
                        d.zout_extra = d.zout + 4 * sb_S[L] * d.zout_slab_stride
                        _lib.call('tfep_inverse_block', ctypes.byref(d), stream)
                        # the super-block's new units as split rows, for the GEMMs of the super-blocks to come
                        if xs is not None and tb['in_range'][1] > tb['in_range'][0]:
                            g0 = tb['in_range'][0] // 8 * 8
                            g1 = min(ops.round_up(tb['in_range'][1], 8), xpad.shape[1] // 8 * 8)
                            ops.split_columns_scaled(xpad, g0, g1 - g0, xs, xs_inv)
                        lo, hi = tb['unit_range'][L - 1]
                        if hi > lo:
                            g0 = lo // 8 * 8
                            ops.split_columns_scaled(h[L - 1], g0, hi - g0, hs, hs_inv)
                        for l_next, (hsl, hsl_inv, _, _) in hs_hidden.items():
                            lo, hi = tb['unit_range'][l_next - 1]
                            if hi > lo:
                                g0 = lo // 8 * 8
                                ops.split_columns_scaled(h[l_next - 1], g0, hi - g0, hsl, hsl_inv)
                    return x, ldj
            self.last_inverse_schedule = 'block_by_block'
            for i_blk, blk in enumerate(bp['blocks']):
                # ---- contribution of all earlier degrees to the whole block, once
                ow = blk['out_wide']
                if fused is not None:
                    # ---- the block's own degrees: wide GEMMs into compact slabs, then one kernel, one thread per sample
                    fb = blk['fused']
                    par = i_blk & 1 if look else 0
                    z, zout = zs[par], zouts[par]
                    if use_sb:
                        sup = supers[blk['sb']]
                        if blk['sb_first']:
                            super_gemms(sup)
                        block_gemms_sb(blk, sup)
                    elif look:
                        if old_done is not None:
                            main.wait_event(old_done)
                        wide_gemms(blk, par, 'kr_new')
                    else:
                        wide_gemms(blk, par, 'kr')
                    w0 = fb['wide0']                           # layer 0: every feature of an earlier block
                    if w0 is not None and use_sb:
                        launch(xpad, packs[0][0], packs[0][1], w0, z[0], 0, act=0, k_split=S)
                    if w0 is not None and look and (i_blk == 0 or not w0['look']):
                        launch(xpad, packs[0][0], packs[0][1], w0, z[0], 0, act=0, k_split=S)      # whole, no look-ahead
                    extra = {wd['layer']: int(look and wd['has_new']) for wd in blk['wide'] + [ow]}
                    if w0 is not None:
                        extra[0] = int(look and i_blk > 0 and w0['look'] and w0['has_new'])
                    for wd in blk['wide'] + ([w0] if w0 is not None else []):
                        d.z[wd['layer']] = z[wd['layer']].data_ptr() - 4 * wd['row0']   # the kernel indexes by packed row
                    for l in range(L):
                        d.z_slabs[l] = S + extra.get(l, 0)
                    d.zout, d.zout_slabs = zout.data_ptr() - 4 * ow['row0'], S_out + extra[L]
                    if use_sb:
                        # the kernel indexes by packed row: the buffers' column 0 is the super-block's first row
                        for wd in blk['wide']:
                            l = wd['layer']
                            sd = [w for w in sup['wide'] if w['layer'] == l][0]
                            d.z[l] = sb_buf[l].data_ptr() - 4 * sd['row0']
                            d.ldz[l], d.z_slab_stride[l] = sb_buf[l].shape[-1], B * sb_buf[l].shape[-1]
                            d.z_slabs[l] = sb_S[l] + int(wd['has_sbnew'])
                        d.zout = sb_buf[L].data_ptr() - 4 * sup['out_wide']['row0']
                        d.ldzout, d.zout_slab_stride = sb_buf[L].shape[-1], B * sb_buf[L].shape[-1]
                        d.zout_slabs = sb_S[L] + int(ow['has_sbnew'])
                else:
                    for wd in blk['wide']:
                        l = wd['layer']
                        launch(h[l - 1], packs[l][0], packs[l][1], wd, z[l], wd['row0'], act=0)
                    launch(h[L - 1], w_out, b_out, ow, zout, ow['row0'], act=0, wide=ow['n_rows'] > 4 * narrow)
                if fused is not None:
                    d.n_steps = fb['n_steps']
                    d.steps, d.feat_cols, d.feat_sel = fb['steps'].data_ptr(), fb['cols'].data_ptr(), fb['sel'].data_ptr()
                    d.feat_in, d.feat_periodic, d.in_cols = fb['feat_in'].data_ptr(), fb['feat_per'].data_ptr(), fb['in_cols'].data_ptr()
                    for l in range(L):
                        d.cache_col0[l], d.cache_n_old[l] = fb['c0'][l], fb['n_old'][l]
                    if look:
                        # everything up to block i_blk - 1 is complete here.  The event is recorded right before the block
                        # kernel so that the kernel (all of a CU's LDS per workgroup) is dispatched first and the
                        # look-ahead GEMMs fill the CUs it leaves free, not the other way round.
                        ready = torch.cuda.Event()
                        ready.record(main)
                    _lib.call('tfep_inverse_block', ctypes.byref(d), stream)
                    if look:
                        old_done = None
                        if i_blk + 1 < len(bp['blocks']):
                            with torch.cuda.stream(side):
                                side.wait_event(ready)
                                wide_gemms(bp['blocks'][i_blk + 1], 1 - par, 'kr_old')
                                old_done = torch.cuda.Event()
                                old_done.record(side)
                    if hs is not None:                      # the block's new units of the last hidden layer, as split rows
                        lo, hi = fb['unit_range'][L - 1]
                        if hi > lo:
                            g0 = lo // 8 * 8
                            ops.split_columns_scaled(h[L - 1], g0, hi - g0, hs, hs_inv)
                        for l_next, (hsl, hsl_inv, _, _) in hs_hidden.items():      # ... and of the panels below it
                            lo, hi = fb['unit_range'][l_next - 1]
                            if hi > lo:
                                g0 = lo // 8 * 8
                                ops.split_columns_scaled(h[l_next - 1], g0, hi - g0, hsl, hsl_inv)
                    continue
                # ---- the block's own degrees, one after the other
                for st in blk['steps']:
                    for hd in st['hidden']:
                        l = hd['layer']
                        if l == 0:
                            launch(xpad, packs[0][0], packs[0][1], hd, h[0], hd['row0'], act=1)
                        else:   # bias is already inside z
                            launch(h[l - 1], packs[l][0], None, hd, h[l], hd['row0'], act=1, pre=z[l], pre_col0=hd['row0'])
                    if st['n_d'] == 0:
                        continue
                    self._step_tables(st, dev)
                    od = st['out']
                    launch(h[L - 1], w_out, None, od, zout, od['row0'], act=0, pre=zout, pre_col0=od['row0'])
                    y_d = ops.gather_columns(y_tr, st['sel'])
                    par = zout[:, od['row0']:od['row0'] + od['n_rows']]
                    kind, sub = st['sub']
                    if kind == 'spline':
                        x_d, _ = ops.spline(y_d, par, sub, inverse=True, log_det_J=ldj)
                    elif kind == 'moebius':
                        x_d, _ = ops.moebius(y_d, par, sub.dimension, sub.max_radius, sub.unit_sphere, inverse=True,
                                             log_det_J=ldj)
                    elif kind == 'shift':       # (a member of a mixed transformer, no periodic features) log-det 0
                        x_d, _ = ops.volume_preserving_shift(y_d, par, inverse=True)
                    else:
                        x_d, _ = ops.affine(y_d, par, inverse=True, log_det_J=ldj)
                    ops.scatter_columns(x_d, st['cols'], x)
                    self._scatter_inputs(xpad, x_d, st['inputs'])
        return x, ldj

    def get_transformer_parameters(self, x: torch.Tensor) -> torch.Tensor:
        """Run the conditioner (reference autoregressive.py:231-247)."""
        if len(self._conditioner_indices) > 0:
            x = ops.gather_columns(x, self._tables(x.device)['cond'])
        if isinstance(self._conditioner, MADE):
            return self._conditioner(x, split=self._use_split_gemm(x.shape[0]))
        return self._conditioner(x)


#: While a HIP graph is being captured the range guard cannot read its flag back; with a dict here it adds the flag to the device
#: counter ``capture_flags['count']`` instead (created -- cleared by a captured fill kernel, so by every replay -- at the first
#: guarded layer call, ``capture_flags['calls']`` counts them), for the owner of the graph to read after a replay
#: (``graphs.GraphedFlow``).
capture_flags = None


#: ``SequentialFlow.forward`` defers the guards of its layers: with a list here a guarded layer call leaves ``(layer, device
#: counter)`` in it and runs on the split kernels without waiting; the sequence reads all the counters at its end -- ONE host
#: synchronisation per flow call instead of one per layer -- and repeats the call from the first flagged layer on.
deferred_flags = None


def _flag_for_capture(x):
    if capture_flags is not None:
        with torch.no_grad():
            capture_flags['count'] = ops.range_flag_device([ops.column_absmax(x.detach())], bits=19, count=capture_flags.get('count'))
        capture_flags['calls'] = capture_flags.get('calls', 0) + 1


class _RangeGuard:
    def __init__(self, layer, x, force=None):
        self.layer, self.x, self.force = layer, x, force

    def __enter__(self):
        layer = self.layer
        self.prev = layer._guard_exact
        if self.force is not None:                      # the backward of a guarded forward: same arithmetic
            layer._guard_exact = bool(self.force)
            return self
        layer._guard_exact = False
        on = layer.split_guard if layer.split_guard is not None else os.environ.get('TFEP_SPLIT_GUARD', '1') != '0'
        x = self.x
        if not on or layer.split_gemm is not None or x.shape[0] == 0 or not layer._use_split_gemm(x.shape[0]):
            return self
        if torch.cuda.is_current_stream_capturing():
            _flag_for_capture(x)                        # (the captured call itself runs on the split kernels)
            return self
        if deferred_flags is not None:
            with torch.no_grad():
                deferred_flags.append((layer, ops.range_flag_device([ops.column_absmax(x.detach())], bits=19)))
            return self
        with torch.no_grad():
            # per-FEATURE magnitudes over the batch (a single small value in one row is harmless -- every unit adds an
            # exact fp32 bias and sees other features; what the per-row scale cannot carry is a feature that is small in
            # EVERY row next to one that is large in every row)
            col_max = ops.column_absmax(x.detach())          # max |x| per feature: one pass at HBM rate, no temporary
            n_x = ops.range_flag([col_max], bits=19)
        layer.last_split_guard = dict(feature_scales_out_of_range=bool(n_x), exact=bool(n_x))
        if n_x:
            layer._guard_exact = True
            drop = getattr(layer._conditioner, 'drop_packed_ahead', None)
            if drop is not None:
                drop()                                  # split weights packed ahead for this call: not used
            if not layer.__dict__.get('_guard_warned'):
                layer.__dict__['_guard_warned'] = True
                import warnings
                warnings.warn('tfep_amd: the input features differ in scale by more than 2^19 (largest magnitude per feature over '
                              'the batch): more than the split-f16 GEMMs carry at fp32 accuracy with one scale per row; this '
                              'layer call runs on the exact-fp32 MFMA kernels (3x slower).  layer.split_gemm = True / False '
                              'pins the arithmetic.')
        return self

    def __exit__(self, *a):
        self.layer._guard_exact = self.prev
        return False


class _null_context:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
