"""MADE conditioner with the ``tfep.nn.conditioners.made`` API on gfx950 kernels.

Mirrors reference ``tfep/nn/conditioners/made.py``: ``generate_degrees`` (:32-145) and
``MADE`` (:152-434) -- same constructor, ``state_dict`` keys (``layers.{0,2,4,..}.{bias,
weight_g, weight_v | weight, mask}``), ``set_output`` and ``n_parameters``.

Execution plan (built lazily per device, host-side integer work only):
  * hidden units are SORTED BY DEGREE (a permutation of hidden units leaves the network
    function unchanged); with sorted units every mask of the network becomes block lower
    triangular, so the GEMM skips ~half of its k-tiles through per-column-tile k-ranges
    computed from the actual mask buffers (``tfep_mask_k_ranges``);
  * weights are re-packed (weight norm + mask + permutation + zero padding) by
    ``tfep_masked_weight_prepare`` on EVERY forward, like the reference's pre-hook;
  * three launches of the fp32-MFMA GEMM, the ELU fused in the first two.
"""
from typing import Literal, Optional, Sequence, Union

import os

import numpy as np

import torch

from ... import ops
from ...utils.misc import ensure_tensor_sequence
from .. import masked
from .conditioner import Conditioner


# =============================================================================
# UTILS
# =============================================================================

def _round_robin(x, length, err_msg=None):
    """Tile ``x`` to ``length`` elements (reference made.py:441-461)."""
    n_rounds, n_rem = divmod(length, len(x))
    if n_rounds == 0:
        if err_msg is None:
            err_msg = f'Length {length} is smaller than the array (len={len(x)}).'
        raise ValueError(err_msg)
    out = x.repeat(n_rounds)
    if n_rem != 0:
        out = torch.cat([out, x[:n_rem]])
    return out


def generate_degrees(
        n_features: int,
        order: Literal['ascending', 'descending', 'random'] = 'ascending',
        max_value: Optional[int] = None,
        conditioning_indices: Optional[Sequence[int]] = None,
        repeats: Union[int, Sequence[int]] = 1,
) -> torch.Tensor:
    """Generate node degrees for MADE layers (reference made.py:32-145).

    Degrees run from 0 to ``max_value``; conditioning features get -1.

    >>> generate_degrees(n_features=3).tolist()
    [0, 1, 2]
    >>> generate_degrees(7, order='descending', max_value=2).tolist()
    [2, 1, 0, 2, 1, 0, 2]
    >>> generate_degrees(7, max_value=2, conditioning_indices=[0, 2, 3]).tolist()
    [-1, 0, -1, -1, 1, 2, 0]
    >>> generate_degrees(7, repeats=[1, 3, 2], conditioning_indices=[2]).tolist()
    [0, 1, -1, 1, 1, 2, 2]
    """
    n_free = n_features
    if conditioning_indices is not None:
        n_free -= len(conditioning_indices)

    if max_value is None:
        try:
            max_value = len(repeats) - 1
        except TypeError:
            max_value = int(np.ceil(n_free / repeats)) - 1

    if order == 'ascending':
        degrees = torch.arange(max_value + 1)
    elif order == 'descending':
        degrees = torch.arange(max_value, -1, -1)
    elif order == 'random':
        degrees = torch.randperm(max_value + 1)
    else:
        raise ValueError("Accepted string values for 'order' "
                         "are 'ascending', 'descending', and 'random'.")

    repeats = ensure_tensor_sequence(repeats, dtype=int)
    degrees = torch.repeat_interleave(degrees, repeats)[:n_free]
    degrees = _round_robin(degrees, length=n_free)

    if conditioning_indices is not None:
        try:
            conditioning_indices = conditioning_indices.tolist()
        except AttributeError:
            pass
        cond = set(conditioning_indices)
        free_idx = [i for i in range(n_features) if i not in cond]
        out = torch.empty(n_features, dtype=degrees.dtype)
        out[list(conditioning_indices)] = -1
        out[free_idx] = degrees
        degrees = out
    return degrees


# =============================================================================
# MADE
# =============================================================================

class MADE(Conditioner):
    """Masked autoencoder conditioner: ``[MaskedLinear, ELU] * n_hidden + MaskedLinear``.

    Arguments as reference made.py:246-284.  Output column ``p*D + f`` is parameter ``p``
    of feature ``f`` when ``degrees_out = transformer.get_degrees_out(degrees)``.
    """

    def __init__(
            self,
            degrees_in: Sequence[int],
            degrees_out: Sequence[int],
            hidden_layers: Union[int, Sequence[int], Sequence[Sequence[int]]] = 2,
            weight_norm: bool = True,
    ):
        super().__init__()
        degrees_in = ensure_tensor_sequence(degrees_in, dtype=int)
        degrees_out = ensure_tensor_sequence(degrees_out, dtype=int)
        degrees_hidden = self._get_degrees_hidden(degrees_in, degrees_out, hidden_layers)
        n_hidden = len(degrees_hidden)

        layers = []
        prev = degrees_in
        for layer_idx in range(n_hidden + 1):
            is_output = layer_idx == n_hidden
            cur = degrees_out if is_output else degrees_hidden[layer_idx]
            # hidden layers '>=', output layer strict '>' (reference made.py:308-309)
            mask = masked.create_autoregressive_mask(prev, cur, strictly_less=is_output, transpose=True)
            lin = masked.MaskedLinear(in_features=len(prev), out_features=len(cur), bias=True, mask=mask)
            if weight_norm:
                lin = masked.masked_weight_norm(lin, name='weight')
            layers.extend([lin, torch.nn.ELU()])
            prev = cur
        layers.pop()
        self.layers = torch.nn.Sequential(*layers)

        # Host-side copy of the degrees (drives the degree sort of the execution plan).
        self._degrees = [d.detach().cpu().clone() for d in (degrees_in, *degrees_hidden, degrees_out)]
        self._degrees_stale = False    # set when the mask buffers are replaced (load_state_dict, in-place edits)
        self._degrees_ok = True        # False: the masks are not those of any degree assignment we can recover
        self._plans = {}
        self._frozen = False
        #: Keep the packed (weight-normed, masked, permuted, split) weights across forwards while no parameter or mask
        #: has been updated: ``Tensor._version`` + storage address, and a strided checksum for writes through ``.data``
        #: that the version cannot see (``_keep_packed``); default off: re-pack on every forward like the reference's
        #: pre-hook (masked.py:397-398).  SURVEY.md section 8(b) sanctions caches invalidated by parameter version.
        self.cache_packed_weights = False
        #: Let the choice between split-f16 and exact-fp32 GEMMs also depend on the batch size (see ``split_worthwhile``);
        #: ``TFEP_SPLIT_BY_BATCH=0`` (or False) pins small conditioners to the exact-fp32 kernels at every batch size.
        self.split_by_batch = os.environ.get('TFEP_SPLIT_BY_BATCH', '1') != '0'
        self._packed_ahead = None      # split weights packed on a side stream for the next forward (prepack_split_async)

    # ------------------------------------------------------------------ reference API
    @property
    def dimension_in(self) -> int:
        return self.layers[0].in_features

    @property
    def dimension_out(self) -> int:
        return self.layers[-1].out_features

    @property
    def dimensions_hidden(self) -> torch.Tensor:
        return torch.tensor([l.out_features for l in self.layers[:-1:2]])

    @property
    def weight_norm(self):
        return self.layers[-1].has_weight_norm

    def n_parameters(self) -> int:
        """The total number of (unmasked) parameters."""
        return sum(l.n_parameters() for l in self.layers[::2])

    def set_output(self, output: torch.Tensor):
        """Make the conditioner return ``output`` for any input (reference made.py:358-364)."""
        last = self.layers[-1]
        if self.weight_norm:
            last.weight_g.data.fill_(0.0)
        else:
            last._parameters['weight'].data.fill_(0.0)
        last.bias.data = output.to(last.bias.data)

    @classmethod
    def _get_degrees_hidden(cls, degrees_in, degrees_out, hidden_layers):
        """Degrees of the hidden nodes (reference made.py:366-434)."""
        try:
            hidden_layers = hidden_layers.tolist()
        except AttributeError:
            pass
        max_degree_out = degrees_out.max()
        relevant = degrees_in < max_degree_out

        if isinstance(hidden_layers, int):
            n_rel = int(relevant.sum())
            n_out = len(degrees_out)
            width = max(int(np.ceil((n_rel * n_out) ** 0.5)), n_rel)
            hidden_layers = [width] * hidden_layers

        if isinstance(hidden_layers[0], int):
            motif = degrees_in[relevant]
            return [
                _round_robin(motif, width, err_msg=(
                    f'Hidden layer {idx} is too small for the number'
                    ' of input features. Increase the size of the layer or'
                    ' explicitly pass the degrees for the hidden layers.'))
                for idx, width in enumerate(hidden_layers)
            ]

        degrees_hidden = [ensure_tensor_sequence(x) for x in hidden_layers]
        for idx, deg in enumerate(degrees_hidden):
            if torch.any(deg >= max_degree_out):
                raise ValueError(f'The {idx}-th hidden layer contain '
                                 'nodes with degrees that will be ignored '
                                 'by the output layer.')
        return degrees_hidden

    # ------------------------------------------------------------------ execution plan
    def _linears(self):
        return list(self.layers[::2])

    def invalidate_plan(self):
        """Drop cached permutations / k-ranges (call after replacing a ``mask`` buffer)."""
        self._plans = {}
        self._packed_ahead = None
        self._degrees_stale = True

    def _load_from_state_dict(self, *args, **kwargs):
        # A checkpoint replaces the mask buffers (e.g. one saved from a model built with another random degree order):
        # everything derived from the degrees -- the host copy, the plans, packed weights -- is re-derived from the
        # loaded buffers on the next use (reference: every quantity derives from the buffers, made.py:286-329).
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_plan()

    def _mask_versions(self):
        return tuple(lin.mask._version for lin in self._linears())

    def _masks_of(self, degrees):
        """True when ``degrees`` (one tensor per layer, inputs first) reproduce every mask buffer."""
        lins = self._linears()
        if len(degrees) != len(lins) + 1:
            return False
        for li, lin in enumerate(lins):
            d_in, d_out = degrees[li].to(lin.mask.device), degrees[li + 1].to(lin.mask.device)
            if (len(d_out), len(d_in)) != tuple(lin.mask.shape):
                return False
            strict = li == len(lins) - 1
            for r0 in range(0, len(d_out), 4096):                       # row chunks: the cfg2 masks are 4.5 GB
                a = d_out[r0:r0 + 4096, None]
                m = (a > d_in[None, :]) if strict else (a >= d_in[None, :])
                if not bool(torch.equal(m, lin.mask[r0:r0 + 4096] != 0)):
                    return False
        return True

    def _sync_degrees(self, degrees_in=None):
        """Make the host copy of the degrees agree with the (possibly just loaded) mask buffers.

        ``degrees_in``: the input degrees when the caller knows them (``AutoregressiveFlow`` derives them from its
        ``_inverse_masks`` buffer).  Hidden and output degrees are re-derived from the masks: a unit's degree is the
        largest degree it is connected to (+1 for the strict output layer) -- exact for MADE, whose hidden degrees
        are drawn from the input degrees (made.py:388-422).  If no assignment reproduces the masks the degree-based
        shortcuts (blocked inverse) are switched off; the GEMMs always take their k-ranges from the actual masks."""
        if not self._degrees_stale:
            return
        self._degrees_stale = False
        self._plans = {}
        self._packed_ahead = None
        if degrees_in is not None:
            degrees_in = ensure_tensor_sequence(degrees_in, dtype=int).detach().cpu()
        current = list(self._degrees)
        if degrees_in is not None and len(degrees_in) == len(current[0]):
            current[0] = degrees_in
        with torch.no_grad():
            if self._masks_of(current):
                self._degrees, self._degrees_ok = [d.clone() for d in current], True
                return
            lins = self._linears()
            derived = [current[0]]
            for li, lin in enumerate(lins):
                prev = derived[-1].to(lin.mask.device)
                if len(prev) != lin.mask.shape[1]:
                    self._degrees_ok = False
                    return
                low = int(prev.min()) - 1
                rows = []
                for r0 in range(0, lin.mask.shape[0], 4096):
                    m = lin.mask[r0:r0 + 4096] != 0
                    rows.append(torch.where(m, prev[None, :], torch.full_like(prev[None, :], low)).max(dim=1).values)
                d = torch.cat(rows)
                if li == len(lins) - 1:
                    d = d + 1
                derived.append(d.cpu())
            self._degrees_ok = self._masks_of(derived)
            if self._degrees_ok:
                self._degrees = derived

    def _apply(self, fn, *args, **kwargs):
        self._plans = {}
        self._packed_ahead = None
        return super()._apply(fn, *args, **kwargs)

    def plan(self, device):
        """Per-device plan: hidden-unit permutations, padded sizes, k-ranges, work buffers."""
        key = str(device)
        cached = self._plans.get(key)
        if cached is not None and cached['mask_versions'] != self._mask_versions():
            self.invalidate_plan()                      # a mask was edited in place
            cached = None
        self._sync_degrees()
        if cached is not None:
            return cached
        tm, tn, tk = ops.tile_sizes()
        lins = self._linears()
        n_lin = len(lins)
        plan = {'row_of_out': [], 'col_of_in': [], 'in_of_col': [], 'n_pad': [], 'k_pad': [], 'k_ranges': [], 'tile_order': [],
                'w': [None] * n_lin, 'bias': [None] * n_lin, 'mask_versions': self._mask_versions()}
        col_of_in = None
        for li, lin in enumerate(lins):
            is_out = li == n_lin - 1
            k_pad = ops.round_up(lin.in_features, tk)
            if is_out:
                row_of_out = None
                n_pad = lin.out_features
            else:
                if self._degrees_ok:
                    deg = self._degrees[li + 1].cpu()
                else:       # masks of unknown origin: fewer connections first (any order is correct, see k_ranges below)
                    deg = torch.count_nonzero(lin.mask, dim=1).cpu()
                order = torch.argsort(deg, stable=True)              # packed position -> hidden unit
                row = torch.empty_like(order)
                row[order] = torch.arange(len(order), device='cpu')                # hidden unit -> packed position
                row_of_out = row.to(device=device, dtype=torch.int32)
                n_pad = ops.round_up(lin.out_features, tk)
            plan['row_of_out'].append(row_of_out)
            plan['col_of_in'].append(col_of_in)
            if col_of_in is None:
                plan['in_of_col'].append(None)
            else:                                                     # packed column -> input unit
                inv = torch.empty_like(col_of_in)
                inv[col_of_in.long()] = torch.arange(len(col_of_in), device=device, dtype=torch.int32)
                plan['in_of_col'].append(inv)
            plan['n_pad'].append(n_pad)
            plan['k_pad'].append(k_pad)
            n_tiles = (n_pad + tn - 1) // tn
            plan['k_ranges'].append(ops.mask_k_ranges(lin.mask, tn, n_tiles, k_pad, row_of_out, col_of_in))
            plan['tile_order'].append(ops.heavy_first_order(plan['k_ranges'][-1]))
            col_of_in = row_of_out
        self._plans[key] = plan
        return plan

    def _pack_layer(self, plan, li, lin, row_of_out=None, n_rows=None):
        """Weight norm + mask + permutation + padding of layer ``li`` (every forward)."""
        row_of_out = plan['row_of_out'][li] if row_of_out is None else row_of_out
        n_rows = plan['n_pad'][li] if n_rows is None else n_rows
        if lin.has_weight_norm:
            v, g = lin.weight_v.detach(), lin.weight_g.detach()
        else:
            v, g = lin._parameters['weight'].detach(), None
        # one buffer per (layer, row order): zeroed once -- the kernel rewrites every mapped entry and never touches the
        # padding, so later packs skip the clear (a 4.5 GB write for the cfg2 output layer)
        key = ('w', li, n_rows, None if row_of_out is None else row_of_out.data_ptr())
        ckey = ('packed',) + key[1:]
        keep = self._keep_packed(plan)
        if keep and ckey in plan:
            return plan[ckey]
        entry = plan.get(key)
        if entry is None:
            # (the entry keeps row_of_out alive: its address, part of the key, cannot be recycled for another mapping)
            entry = plan[key] = (ops.zeros(n_rows, plan['k_pad'][li], dtype=torch.float32, device=v.device), row_of_out)
        buf = entry[0]
        ops.masked_weight_prepare(v, g, lin.mask, row_of_out, plan['col_of_in'][li], n_rows, plan['k_pad'][li], out=buf,
                                  col_cut=self._mask_prefix_cuts(plan, li, lin), clear=False, in_of_col=plan['in_of_col'][li])
        bias = ops.zeros(1, n_rows, dtype=torch.float32, device=v.device)
        if row_of_out is None:
            bias[0, :lin.out_features] = lin.bias.detach()
        else:
            ops.scatter_columns(lin.bias.detach()[None, :], row_of_out, bias)
        if keep:
            plan[ckey] = (buf, bias[0])
        return buf, bias[0]

    def _pack_bias(self, lin, row_of_out, n_rows):
        bias = ops.zeros(1, n_rows, dtype=torch.float32, device=lin.bias.device)
        if row_of_out is None:
            bias[0, :lin.out_features] = lin.bias.detach()
        else:
            ops.scatter_columns(lin.bias.detach()[None, :], row_of_out, bias)
        return bias[0]

    def _pack_layer_both(self, plan, li, lin, row_of_out=None, n_rows=None):
        """Fill the caches of ``_pack_layer`` (fp32) AND ``_pack_layer_split`` for layer ``li`` from one pass over the
        parameters (``tfep_masked_weight_prepare_split_both``) -- inside ``frozen_weights()`` / with ``cache_packed_weights``,
        for a layer whose mask rows are prefixes and whose rows hold 8192 .. 16 384 weights; otherwise nothing happens and the
        two methods pack on their own.  The blocked inverse wants both forms of every layer."""
        if not self._keep_packed(plan) or not (8192 <= lin.in_features <= 16384):
            return False
        row_of_out = plan['row_of_out'][li] if row_of_out is None else row_of_out
        n_rows = plan['n_pad'][li] if n_rows is None else n_rows
        rptr = None if row_of_out is None else row_of_out.data_ptr()
        ck32, cks = ('packed', li, n_rows, rptr), ('packed_split', li, n_rows, rptr)
        if ck32 in plan and cks in plan:
            return True
        cut = self._mask_prefix_cuts(plan, li, lin)
        in_of_col = plan['in_of_col'][li]
        if cut is None or (in_of_col is None and plan['col_of_in'][li] is not None) or os.environ.get('TFEP_PACK_BOTH', '1') == '0':
            return False
        if lin.has_weight_norm:
            v, g = lin.weight_v.detach(), lin.weight_g.detach()
        else:
            v, g = lin._parameters['weight'].detach(), None
        k32, ks = ('w', li, n_rows, rptr), ('ws', li, n_rows, rptr)
        e32 = plan.get(k32)
        if e32 is None:
            e32 = plan[k32] = (ops.zeros(n_rows, plan['k_pad'][li], dtype=torch.float32, device=v.device), row_of_out)
        es = plan.get(ks)
        if es is None:
            es = plan[ks] = (ops.zeros(n_rows, plan['k_pad'][li], dtype=torch.float32, device=v.device),
                             ops.zeros(4, dtype=torch.float32, device=v.device), row_of_out)
        v_c = v.contiguous()
        g_c = None if g is None else g.contiguous()
        ops.call('tfep_masked_weight_prepare_split_both', ops.ptr(v_c), ops.ptr(g_c), v.shape[0], v.shape[1], ops.ptr(row_of_out),
                  ops.ptr(in_of_col), ops.ptr(cut), ops.ptr(es[0]), es[0].shape[1], ops.ptr(e32[0]), e32[0].shape[1],
                  plan['k_pad'][li], ops.ptr(es[1]), ops.stream_of(v))
        bias = self._pack_bias(lin, row_of_out, n_rows)
        plan[ck32] = (e32[0], bias)
        plan[cks] = (es[0], es[1], bias, ops.abs_reduce(bias.reshape(1, -1), 'row_max'))
        return True

    def _pack_layer_split(self, plan, li, lin, row_of_out=None, n_rows=None):
        """Weight norm + mask + permutation + padding of layer ``li`` written directly as split-f16 rows (one scale
        for the matrix).  Returns ``(w_split, w_inv_scale, bias)``."""
        row_of_out = plan['row_of_out'][li] if row_of_out is None else row_of_out
        n_rows = plan['n_pad'][li] if n_rows is None else n_rows
        ckey = ('packed_split', li, n_rows, None if row_of_out is None else row_of_out.data_ptr())
        keep = self._keep_packed(plan)
        if keep and ckey in plan:
            return plan[ckey]
        ahead = self._packed_ahead
        if ahead is not None and ahead['versions'] != self._param_versions():
            ahead = self._packed_ahead = None                     # parameters changed since: pack again
        if ahead is not None and (li, n_rows) in ahead['items']:
            # packed on the side stream while the previous layer of the flow was computing (prepack_split_async)
            if not ahead['joined']:
                torch.cuda.current_stream(ahead['device']).wait_event(ahead['event'])
                ahead['joined'] = True
            res = ahead['items'].pop((li, n_rows))
            for t in res[2:]:
                t.record_stream(torch.cuda.current_stream(ahead['device']))     # small tensors allocated on the side stream
            return res
        if lin.has_weight_norm:
            v, g = lin.weight_v.detach(), lin.weight_g.detach()
        else:
            v, g = lin._parameters['weight'].detach(), None
        key = ('ws', li, n_rows, None if row_of_out is None else row_of_out.data_ptr())
        buf = plan.get(key)
        if buf is None:
            # zero once: the kernel writes the real rows only, padding rows / columns stay zero for good
            # (row_of_out rides along so that its address, part of the key, cannot be recycled for another mapping)
            buf = (ops.zeros(n_rows, plan['k_pad'][li], dtype=torch.float32, device=v.device),
                   ops.zeros(4, dtype=torch.float32, device=v.device), row_of_out)
            plan[key] = buf
        in_of_col = plan['in_of_col'][li]
        ops.masked_weight_prepare_split(v, g, lin.mask, row_of_out, in_of_col, buf[0], buf[1],
                                        col_cut=self._mask_prefix_cuts(plan, li, lin))
        bias = self._pack_bias(lin, row_of_out, n_rows)
        res = (buf[0], buf[1], bias, ops.abs_reduce(bias.reshape(1, -1), 'row_max'))
        if keep:
            plan[ckey] = res
        return res

    def _keep_packed(self, plan):
        """True when packed weights may be kept in (and served from) ``plan``: inside ``frozen_weights()``, or with
        ``cache_packed_weights`` while every parameter / mask still has the version the cache was filled at."""
        if self._frozen:
            return True
        if not self.cache_packed_weights or torch.cuda.is_current_stream_capturing():
            return False
        versions = self._param_versions()
        # ``Tensor._version`` does not see in-place writes through ``.data`` (legacy loops: ``p.data.add_(...)``): a
        # strided device checksum of every parameter (every 1021st entry: ~1 M reads per cfg2 layer, one host comparison)
        # catches any update that touches a tensor broadly -- every optimiser step does.  A write to a SINGLE entry
        # through ``.data`` is the one thing that can still go unseen; ``invalidate_plan()`` is the explicit remedy.
        fingerprint = self.__dict__.get('_fp_memo')          # one checksum per layer call (begin_call), not one per linear
        if fingerprint is None:
            fingerprint = self._param_fingerprint()
            self.__dict__['_fp_memo'] = fingerprint
        stale = plan.get('packed_versions') != versions
        if not stale:
            # the device comparison (a host sync) once per plan and call, not once per linear and pack variant: memoised on the
            # identity of this call's checksum tensor
            if plan.get('packed_checked') is fingerprint:
                return True
            old = plan.get('packed_fingerprint')
            stale = old is None or not torch.equal(old, fingerprint)
        plan['packed_checked'] = fingerprint
        if stale:
            for k in [k for k in plan if isinstance(k, tuple) and k[0] in ('packed', 'packed_split')]:
                del plan[k]
            plan['packed_versions'] = versions
        plan['packed_fingerprint'] = fingerprint
        return True

    def begin_call(self):
        """Start of one forward / inverse / backward call of the owning layer (or of ``forward`` itself): the parameter
        checksum of ``cache_packed_weights`` is taken at most once per call."""
        self.__dict__['_fp_memo'] = None

    def _param_fingerprint(self):
        """Strided checksums (float64 sums of every 1021st entry, and of the first 64) of every parameter: a device tensor."""
        parts = []
        for lin in self._linears():
            for t in lin.parameters():
                flat = t.detach().reshape(-1)
                parts.append(flat[::1021].sum(dtype=torch.float64))
                parts.append(flat[:64].sum(dtype=torch.float64))
        return torch.stack(parts) if parts else torch.zeros(0)

    def _mask_prefix_cuts(self, plan, li, lin):
        """``col_cut`` of ``tfep_masked_weight_prepare_split`` for layer ``li``, or None.

        The packed columns of a layer are sorted by the degree of their input unit, so a row of an autoregressive mask
        (made.py:308-309: ``deg_out > deg_in`` / ``>=``) is a PREFIX of them: ``mask[o, in_of_col[c]] == (c < cut[o])``.  The
        re-pack then needs no mask at all -- a third of its HBM traffic (4.5 GB per cfg2 output layer and forward).  The
        property is checked against the actual mask buffer, once per mask version; any other mask is read as before."""
        if os.environ.get('TFEP_MASK_PREFIX', '1') == '0':      # A/B switch: always read the mask
            return None
        key = ('col_cut', li)
        cached = plan.get(key)
        version = lin.mask._version
        if cached is not None and cached[0] == version:
            return cached[1]
        if torch.cuda.is_current_stream_capturing():
            return None                                       # (the check synchronises: not inside a graph capture)
        in_of_col = plan['in_of_col'][li]
        gather = None if in_of_col is None else in_of_col.long()
        cuts, prefix = [], True
        with torch.no_grad():
            for r0 in range(0, lin.mask.shape[0], 4096):
                m = lin.mask[r0:r0 + 4096]
                m = m if gather is None else m[:, gather]
                if not bool((((m == 0) | (m == 1)).all() & (m[:, 1:] <= m[:, :-1]).all()).item()):
                    prefix = False
                    break
                cuts.append(m.sum(dim=1).to(torch.int32))
        cut = torch.cat(cuts).contiguous() if prefix and cuts else None
        plan[key] = (version, cut)
        return cut

    def prepack_split_async(self, device, stream, last=None):
        """Pack every layer's split weights on ``stream`` (ordered after everything already queued on the current
        stream), for the NEXT forward pass of this conditioner: the HBM-bound weight preparation of one flow layer
        then overlaps the (power-bound) GEMMs of the layer before it.  ``last``: ``(row_of_out, n_rows)`` of the output
        layer when the fused path packs it its own way."""
        plan = self.plan(device)
        lins = self._linears()
        self._packed_ahead = None
        if self.cache_packed_weights and not self._frozen:
            return                                 # the cache already holds (or the forward will fill) the packed weights
        stream.wait_stream(torch.cuda.current_stream(device))
        items = {}
        with torch.cuda.stream(stream):
            for li, lin in enumerate(lins):
                if li == len(lins) - 1 and last is not None:
                    items[(li, last[1])] = self._pack_layer_split(plan, li, lin, row_of_out=last[0], n_rows=last[1])
                else:
                    items[(li, plan['n_pad'][li])] = self._pack_layer_split(plan, li, lin)
            event = stream.record_event()
        self._packed_ahead = dict(items=items, event=event, joined=False, device=device, versions=self._param_versions())

    def _param_versions(self):
        """In-place update counters of every tensor the packed weights depend on."""
        return tuple((t._version, t.data_ptr()) for lin in self._linears() for t in (*lin.parameters(), lin.mask))

    def drop_packed_ahead(self):
        self._packed_ahead = None

    def _embed(self, x):
        return x

    def frozen_weights(self):
        """Context manager: pack the weights once and reuse them for every forward inside the
        block (the D sequential passes of the autoregressive inverse share one set of weights)."""
        return _FrozenWeights(self)

    def forward_hidden(self, x, split=False):
        """Run every layer but the last; returns the last hidden activations (zero padded,
        units sorted by degree) and the plan.  ``split``: the GEMMs take split-f16 operands
        (``csrc/split_gemm.hip``); the activations returned are fp32 either way."""
        ops.check_device_tensor(x, 'x')
        x = self._embed(x)
        if x.shape[1] != self.dimension_in:
            raise ValueError(f'expected {self.dimension_in} input features, got {x.shape[1]}')
        plan = self.plan(x.device)
        lins = self._linears()
        h = ops.pad_columns(x, plan['k_pad'][0])
        for li, lin in enumerate(lins[:-1]):
            if split:
                ws, w_inv, b, _ = self._pack_layer_split(plan, li, lin)
                hs, h_inv = ops.split_rows(h, plan['k_pad'][li])
                h = ops.masked_linear_split(hs, h_inv, ws, w_inv, b, plan['n_pad'][li], k_ranges=plan['k_ranges'][li],
                                            act=1, tile_order=plan['tile_order'][li])
            else:
                w, b = self._pack_layer(plan, li, lin)
                h = ops.masked_linear_packed(h, w, b, plan['n_pad'][li], k_ranges=plan['k_ranges'][li], act=1,
                                             tile_order=plan['tile_order'][li])
        return h, plan

    def split_worthwhile(self, batch=None):
        """True when the GEMMs are large enough for the split-f16 kernels to pay for their operand conversions: at
        least 4 M weights; with ``split_by_batch`` (the default) also at least 2^35 weight x row products (a 3 M-weight
        conditioner at batch 131 072, BASELINE cfg4-ii, is GEMM-bound all the same: 20.2 -> 12.7 ms).  That second rule
        makes the ARITHMETIC depend on the batch size: a row pushed through in a small batch equals the same row in a
        large one to the split format's 2^-22, not bit for bit (like the blocked inverse, whose schedule follows the batch
        size too; the reference's BLAS does not promise batch-independent bits either).  ``split_by_batch = False`` /
        ``TFEP_SPLIT_BY_BATCH=0`` restores one arithmetic per conditioner.  Tiny problems are launch bound either way and
        keep the exact-fp32 kernel, which needs no conversions."""
        n = self.__dict__.get('_n_weights')
        if n is None:
            n = self.__dict__['_n_weights'] = sum(lin.mask.numel() for lin in self._linears())
        by_batch = self.split_by_batch and batch is not None and n * int(batch) >= (1 << 35)
        return n >= (1 << 22) or by_batch

    def forward_hidden_split(self, x, split_hint=None):
        """``forward_hidden`` for the split-f16 GEMMs without fp32 intermediates: every hidden layer writes its
        ELU activations directly as split rows (scale from a bound on the row, see ``EPI_ELU_SPLIT``).
        Returns ``(h_split, h_inv_scale, plan)`` of the last hidden layer.  ``split_hint``: ``(rows, inv_scale, version)`` of
        ``x`` already as split rows (the ``_tfep_split`` attribute its producer left on the tensor), used when ``x`` has not
        been written since and is the conditioner's input as it stands (no embedding)."""
        ops.check_device_tensor(x, 'x')
        if split_hint is None:
            split_hint = getattr(x, '_tfep_split', None)
        hint_version = x._version
        x_in = x
        x = self._embed(x)
        if x is not x_in:
            split_hint = None
        if x.shape[1] != self.dimension_in:
            raise ValueError(f'expected {self.dimension_in} input features, got {x.shape[1]}')
        plan = self.plan(x.device)
        lins = self._linears()
        if split_hint is not None and split_hint[2] == hint_version and tuple(split_hint[0].shape) == (x.shape[0], plan['k_pad'][0]):
            hs, h_inv = split_hint[0], split_hint[1]          # the producer of x wrote its split rows too (flows/autoregressive.py)
        else:
            hs, h_inv = ops.split_rows(x, plan['k_pad'][0])
        for li, lin in enumerate(lins[:-1]):
            ws, w_inv, b, bmax = self._pack_layer_split(plan, li, lin)
            if plan['n_pad'][li] != plan['k_pad'][li + 1]:
                raise RuntimeError('hidden layer width and next layer input padding differ')
            hs, h_inv = ops.masked_linear_split(hs, h_inv, ws, w_inv, b, plan['n_pad'][li], k_ranges=plan['k_ranges'][li],
                                                act=1, tile_order=plan['tile_order'][li], split_out=True, bias_absmax=bmax)
        return hs, h_inv, plan

    def forward(self, x, split=None):
        """Transformer parameters ``(..., n_out)`` (reference made.py:355).  ``split``: run the GEMMs on split-f16
        operands (fp32-equivalent, ``csrc/split_gemm.hip``); None = the ``TFEP_SPLIT_GEMM`` default."""
        self.begin_call()
        lead = x.shape[:-1]
        hint = getattr(x, '_tfep_split', None) if x.dim() == 2 else None      # (a reshape makes a new tensor object)
        if hint is not None and hint[2] != x._version:
            hint = None
        x2 = x.reshape(-1, x.shape[-1])
        split = (ops.split_gemm_enabled() and self.split_worthwhile(x2.shape[0])) if split is None else bool(split)
        if split:
            hs, h_inv, plan = self.forward_hidden_split(x2, None if hint is None else (hint[0], hint[1], x2._version))
        else:
            h, plan = self.forward_hidden(x2)
        li = len(plan['n_pad']) - 1
        lin = self.layers[-1]
        if split:
            ws, w_inv, b, _ = self._pack_layer_split(plan, li, lin)
            out = ops.masked_linear_split(hs, h_inv, ws, w_inv, b, lin.out_features, k_ranges=plan['k_ranges'][li],
                                          act=0, tile_order=plan['tile_order'][li])
        else:
            w, b = self._pack_layer(plan, li, lin)
            out = ops.masked_linear_packed(h, w, b, lin.out_features, k_ranges=plan['k_ranges'][li], act=0,
                                           tile_order=plan['tile_order'][li])
        return out.reshape(*lead, lin.out_features)


class _FrozenWeights:
    def __init__(self, made):
        self.made = made

    def _drop(self):
        for plan in self.made._plans.values():
            for k in [k for k in plan if isinstance(k, tuple) and k[0] in ('packed', 'packed_split')]:
                del plan[k]

    def _cached(self):
        # with ``cache_packed_weights`` the packs outlive the block like those of the forward pass: a sampling loop (inverse
        # after inverse on the same weights) then packs once -- 5.7 of 83 ms per inverse of a cfg2 layer at B = 8192
        return self.made.cache_packed_weights and not torch.cuda.is_current_stream_capturing()

    def __enter__(self):
        if self._cached():
            self.made.begin_call()
            for plan in self.made._plans.values():
                self.made._keep_packed(plan)          # (drops what a parameter update has made stale)
        else:
            self._drop()
        self.made._frozen = True
        return self.made

    def __exit__(self, *a):
        self.made._frozen = False
        if not self._cached():
            self._drop()
        return False
