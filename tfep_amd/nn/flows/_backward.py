"""Backward pass of one MAF layer on the HIP kernels (training step, reference app/base.py:780-840).

The reference differentiates the flow with eager autograd plus ``MaskedLinearFunc.backward``
(masked.py:279-302) and the weight-norm gradient hooks (masked.py:401-402, :429).  Here one
``torch.autograd.Function`` per MAF layer wraps the (fused) HIP forward; its backward

  1. recomputes the hidden activations and the transformer parameters for a chunk of the batch
     (so the ``(B, P*D)`` parameter tensor only ever exists for a chunk),
  2. runs the transformer VJP kernel (``tfep_affine_backward`` / ``tfep_spline_backward``),
  3. walks the three masked linears backwards with the SAME fp32-MFMA GEMM kernel:
     ``grad_input = g W`` on a transposed packed weight (ELU derivative fused in the epilogue),
     ``grad_weight += g^T x`` on transposed activations, skipping output tiles that the block-triangular
     mask kills, ``grad_bias += column sums``,
  4. converts the packed weight gradients to ``weight_v`` / ``weight_g`` / ``weight`` gradients
     (``tfep_weight_norm_backward``).

PyTorch's role is the autograd graph between layers, the loss and the optimiser.
"""
import ctypes

import os

import torch

from ... import _lib, ops
from ..conditioners.made import MADE
from ..masked import GRAD_IS_MASKED
from ..embeddings.mafembed import MAFEmbedding, PeriodicEmbedding
from ..transformers.affine import AffineTransformer, VolumePreservingShiftTransformer
from ..transformers.mixed import MixedTransformer
from ..transformers.moebius import MoebiusTransformer
from ..transformers.spline import NeuralSplineTransformer

# bytes of transformer parameters per batch chunk of the backward (TFEP_BACKWARD_CHUNK_GIB, default 8:
# sized for 288 GB of HBM -- fewer, larger chunks mean fewer read-modify-write passes over the gradient of the 4.5 GB
# output-layer weight and longer k in the grad_weight GEMMs; 2 -> 8 GiB: 348 -> 315 ms per cfg2 layer at B=16384)
_CHUNK_BYTES = int(float(os.environ.get('TFEP_BACKWARD_CHUNK_GIB', 8)) * (1 << 30))

# A training forward KEEPS the hidden activations and the transformer parameters of the layer for its backward when the
# parameters take at most this many bytes (TFEP_SAVE_ACTIVATIONS_GIB, default 24; 0 = always recompute).  4.9 GB per cfg2
# layer at B = 16 384 against 288 GB of HBM: storing them costs ~1 ms of writes, recomputing them costs the two largest
# GEMMs of the layer a second time (52 of 303 ms).  The forward then runs the un-fused kernels (GEMM -> parameters in HBM
# -> transformer kernel) on the backward's weight packing, which is shared between the two passes.
_SAVE_BYTES = int(float(os.environ.get('TFEP_SAVE_ACTIVATIONS_GIB', 24)) * (1 << 30))


def supported(layer):
    made = layer._conditioner
    if not isinstance(made, MADE) or len(layer._conditioner_indices) > 0:
        return False
    emb = getattr(made, 'embedding', None)
    if emb is not None and not isinstance(emb, MAFEmbedding):
        return False
    return _transformer_supported(layer._transformer)


def _embedding_params(layer):
    """Trainable parameters of the conditioner's embedding (FlipInvariantEmbedding networks), if any."""
    emb = getattr(layer._conditioner, 'embedding', None)
    return [] if emb is None else list(emb.parameters())


def _transformer_supported(tr):
    if type(tr) in (AffineTransformer, MoebiusTransformer, VolumePreservingShiftTransformer):
        return True
    if type(tr) is NeuralSplineTransformer:
        return True
    if type(tr) is MixedTransformer:
        return all(_transformer_supported(t) for t in tr._transformers)
    return False


def _voidp(t, offset_elems=0):
    return ctypes.c_void_p(t.data_ptr() + 4 * offset_elems)


def transformer_vjp(tr, x, theta, th_off, ld_theta, gy, gl, gtheta, gx, stream, layout=None, order=None):
    """VJP of ``tr.forward(x, theta[:, th_off:th_off + n_par])``: writes the matching block of ``gtheta`` and
    ``gx`` (contiguous (B, D)).  ``theta`` / ``gtheta`` have row stride ``ld_theta``."""
    B, D = x.shape
    dev = x.device
    if type(tr) is MixedTransformer:
        key = str(dev)
        if key not in tr._i32:
            tr._i32[key] = [ind.to(device=dev, dtype=torch.int32) for ind in tr._indices]
        splits = tr.host_splits()
        for sub, ind, a in zip(tr._transformers, tr._i32[key], splits):
            xs, gys = ops.gather_columns(x, ind), ops.gather_columns(gy, ind)
            gxs = torch.empty_like(xs)
            transformer_vjp(sub, xs, theta, th_off + a, ld_theta, gys, gl, gtheta, gxs, stream)
            ops.scatter_columns(gxs, ind, gx)
        return
    th, gth = _voidp(theta, th_off), _voidp(gtheta, th_off)
    # layout: (row stride, stride between parameters, stride between features); default = reference layout.
    # `order`: the features are given in this permutation (slot order), so per-feature constants follow it.
    lay = _lib.ParamLayout(*layout) if layout is not None else _lib.ParamLayout(ld_theta, D, 1)
    if type(tr) is NeuralSplineTransformer:
        cfg = _slot_config(tr, dev, order)
        _lib.call('tfep_spline_backward', _lib.ptr(x), D, th, lay, ctypes.byref(cfg.desc), _lib.ptr(gy), D,
                  _lib.ptr(gl), gth, lay, _lib.ptr(gx), D, B, D, stream)
    elif type(tr) is AffineTransformer:
        _lib.call('tfep_affine_backward', _lib.ptr(x), D, th, lay, _lib.ptr(gy), D, _lib.ptr(gl), gth, lay,
                  _lib.ptr(gx), D, B, D, stream)
    elif type(tr) is MoebiusTransformer:
        _lib.call('tfep_moebius_backward', _lib.ptr(x), D, th, ld_theta, int(tr.dimension), float(tr.max_radius),
                  int(bool(tr.unit_sphere)), 1, _lib.ptr(gy), D, _lib.ptr(gl), gth, ld_theta, _lib.ptr(gx), D, B, D,
                  stream)
    else:   # volume-preserving shift: y = x + b (wrap is piecewise identity), log-det = 0
        _lib.call('tfep_copy_2d', _lib.ptr(gy), D, gth, ld_theta, B, D, stream)
        _lib.call('tfep_copy_2d', _lib.ptr(gy), D, _lib.ptr(gx), D, B, D, stream)


def _slot_config(tr, dev, order):
    """Spline constants of ``tr`` in the feature permutation ``order`` (cached on the transformer)."""
    cfg = tr.config(dev)
    if order is None:
        return cfg
    key = ('slot_cfg', str(dev))
    if tr.__dict__.get('_slot_cfg_key') != (key, id(order)):
        h = tr.host()
        o = order.long()
        tr.__dict__['_slot_cfg'] = ops.SplineConfig(cfg.x0[o], cfg.xf[o], cfg.y0[o], cfg.yf[o], h['n_bins'],
                                                    h['circular'], h['identity'], h['learn_lower'],
                                                    h['learn_upper'], h['min_bin'], h['min_slope'])
        tr.__dict__['_slot_cfg_key'] = (key, id(order))
    return tr.__dict__['_slot_cfg']


def transformer_forward(tr, x, theta, layout, order, stream):
    """``(y, log_det_J)`` of an affine / spline transformer on parameters stored as ``layout`` says, features of ``x``
    in the permutation ``order`` (the counterpart of ``transformer_vjp`` for the activation-saving forward)."""
    B, D = x.shape
    y = torch.empty(B, D, dtype=torch.float32, device=x.device)
    ldj = torch.empty(B, dtype=torch.float32, device=x.device)
    lay = _lib.ParamLayout(*layout)
    if type(tr) is NeuralSplineTransformer:
        cfg = _slot_config(tr, x.device, order)
        _lib.call('tfep_spline_forward', _lib.ptr(x), D, _lib.ptr(theta), lay, ctypes.byref(cfg.desc), _lib.ptr(y), D,
                  _lib.ptr(ldj), 0, B, D, stream)
    else:
        _lib.call('tfep_affine_forward', _lib.ptr(x), D, _lib.ptr(theta), lay, _lib.ptr(y), D, _lib.ptr(ldj), 0, B, D, stream)
    return y, ldj


def trainable_tensors(layer):
    """The conditioner parameters, in a fixed order shared by forward() and backward()."""
    out = []
    for lin in layer._conditioner._linears():
        if lin.has_weight_norm:
            out += [lin.weight_g, lin.weight_v]
        else:
            out += [lin._parameters['weight']]
        out.append(lin.bias)
    return out + _embedding_params(layer)


class MAFLayerFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, layer, x, *params):
        saved = None
        with torch.no_grad():
            if saves_activations(layer, x):
                y, ldj, saved = forward_saving(layer, x)
            else:
                y, ldj = layer._forward_impl(x)
        ctx.layer = layer
        ctx.guard_exact = layer._guard_exact            # the range guard's verdict on this call's data: the backward keeps it
        # the kept activations go through save_for_backward like x: released with the graph / right after backward (as
        # plain attributes of ctx they sat in a reference cycle until Python's garbage collector ran: 6.6 GB per step)
        ctx.n_hidden = -1 if saved is None else len(saved['h'])
        ctx.save_for_backward(x, *(() if saved is None else (*saved['h'], saved['theta'])))
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj):
        layer = ctx.layer
        x, *kept = ctx.saved_tensors
        saved = None if ctx.n_hidden < 0 else dict(h=kept[:ctx.n_hidden], theta=kept[ctx.n_hidden])
        del kept
        from .autoregressive import _RangeGuard
        with torch.no_grad(), _RangeGuard(layer, x, force=ctx.guard_exact):
            # (the first layer of a flow: the data needs no gradient -- its grad_input GEMM and embedding VJP are skipped)
            gx, gparams = layer_backward(layer, x, gy, gldj, saved=saved, need_gx=ctx.needs_input_grad[1])
        # parameters this backward does not hand a gradient to (torch.autograd.grad(loss, [x]), a subset of the
        # parameters): their hooks will not run, so their "already masked" tags must not outlive this call
        for prm, needed in zip(trainable_tensors(layer), ctx.needs_input_grad[2:]):
            if not needed:
                prm.__dict__.pop(GRAD_IS_MASKED, None)
        return (None, gx, *gparams)


class TransformerFunction(torch.autograd.Function):
    """``y, log_det_J = transformer(x, theta)`` on the HIP kernels as one graph node; the backward is the transformer
    VJP kernel (``transformer_vjp``).  Used by ``generic_forward``: everything AROUND the transformer -- a user-supplied
    conditioner, an embedding that is not a ``MAFEmbedding``, a conditioner fed from a subset of the features -- is
    then differentiated by ordinary autograd, as the reference does for the whole layer."""

    @staticmethod
    def forward(ctx, tr, x, theta):
        x, theta = x.detach().contiguous(), theta.detach().contiguous()
        with torch.no_grad():
            y, ldj = tr(x, theta)
        ctx.tr = tr
        ctx.save_for_backward(x, theta)
        return y, ldj

    @staticmethod
    def backward(ctx, gy, gldj):
        x, theta = ctx.saved_tensors
        B, D = x.shape
        f32 = dict(dtype=torch.float32, device=x.device)
        gy = ops.zeros(B, D, **f32) if gy is None else gy.contiguous()
        gl = ops.zeros(B, **f32) if gldj is None else gldj.contiguous()
        gtheta = ops.zeros(*theta.shape, **f32)
        gx = torch.empty(B, D, **f32)
        with torch.no_grad():
            transformer_vjp(ctx.tr, x, theta, 0, theta.shape[1], gy, gl, gtheta, gx, _lib.stream_of(x))
        return None, gx, gtheta


_HIP_TRANSFORMERS = (AffineTransformer, MoebiusTransformer, VolumePreservingShiftTransformer, NeuralSplineTransformer,
                     MixedTransformer)


def generic_supported(layer):
    """Layers whose backward is not the fused HIP one but can still be differentiated: the transformer has a VJP kernel,
    or it is a user-supplied torch module that autograd differentiates by itself.  (A SUBCLASS of one of this package's
    transformers may have changed the map but still runs the HIP forward, which autograd cannot see: not supported.)"""
    tr = layer._transformer
    return _transformer_supported(tr) or not isinstance(tr, _HIP_TRANSFORMERS)


def generic_forward(layer, x):
    """Differentiable forward of a layer that ``supported`` rejects: the conditioner runs under autograd (a MADE through
    its ``MaskedLinear`` modules, reference masked.py:279-302 semantics; anything else as the torch module it is), the
    transformer through ``TransformerFunction``, the index plumbing through the differentiable gather / replace of
    ``flows/partial.py``.  Slower than the fused path (the (B, P D) parameters are materialised and kept for the
    backward) -- the point is that training works for every configuration the reference can train."""
    from .partial import _GatherColumns, _ReplaceColumns
    t = layer._tables(x.device)
    cond_in = _GatherColumns.apply(x, t['cond']) if len(layer._conditioner_indices) > 0 else x
    made = layer._conditioner
    if isinstance(made, MADE):
        theta = made.layers(made._embed(cond_in))
    else:
        theta = made(cond_in)
    tr = layer._transformer
    if _transformer_supported(tr):
        def apply(x_tr):
            return TransformerFunction.apply(tr, x_tr, theta)
    else:                                                    # a user's torch transformer: plain autograd
        def apply(x_tr):
            return tr(x_tr, theta)
    if layer.has_fixed_indices:
        y_tr, ldj = apply(_GatherColumns.apply(x, t['tr']))
        return _ReplaceColumns.apply(x, y_tr, t['tr']), ldj
    return apply(x)


def _elementwise(tr):
    """Transformers whose Jacobian with respect to x is diagonal (one feature in, one feature out)."""
    if type(tr) in (AffineTransformer, NeuralSplineTransformer, VolumePreservingShiftTransformer):
        return True
    return type(tr) is MixedTransformer and all(_elementwise(t) for t in tr._transformers)


class TransformerInverseFunction(torch.autograd.Function):
    """``x, log_det_J = transformer.inverse(y, theta)`` on the HIP kernels as one graph node, for the element-wise
    transformers (affine, RQ spline in every layout, volume-preserving shift, mixed transformers of those).

    The backward needs no kernel of its own.  With y = tau(x; theta) and L = log tau_x (per feature), the inverse returns
    (x, -sum_f L), and dx = (dy - tau_theta dtheta) / tau_x.  For cotangents (gx, gl) of its two outputs:

        u = (gx - gl L_x) / tau_x,        g_y = u,        g_theta = -(tau_theta^T u + gl L_theta)

    The forward VJP kernel called with cotangents (a, b) returns (tau_x a + L_x b, tau_theta^T a + L_theta^T b): (1, 0)
    yields tau_x, (0, gl) yields gl L_x, and (u, gl) yields -g_theta."""

    @staticmethod
    def forward(ctx, tr, y, theta):
        y, theta = y.detach().contiguous(), theta.detach().contiguous()
        with torch.no_grad():
            x, ldj = tr.inverse(y, theta)
        ctx.tr = tr
        ctx.save_for_backward(x, theta)
        return x, ldj

    @staticmethod
    def backward(ctx, gx, gldj):
        x, theta = ctx.saved_tensors
        B, D = x.shape
        f32 = dict(dtype=torch.float32, device=x.device)
        stream = _lib.stream_of(x)
        gx = ops.zeros(B, D, **f32) if gx is None else gx.contiguous()
        gl = ops.zeros(B, **f32) if gldj is None else gldj.contiguous()
        with torch.no_grad():
            scratch = ops.zeros(*theta.shape, **f32)
            tau_x = torch.empty(B, D, **f32)
            transformer_vjp(ctx.tr, x, theta, 0, theta.shape[1], torch.ones(B, D, **f32), ops.zeros(B, **f32), scratch, tau_x, stream)
            lx = torch.empty(B, D, **f32)
            transformer_vjp(ctx.tr, x, theta, 0, theta.shape[1], ops.zeros(B, D, **f32), gl, scratch, lx, stream)
            u = ((gx - lx) / tau_x).contiguous()
            gtheta = ops.zeros(*theta.shape, **f32)
            dump = torch.empty(B, D, **f32)
            transformer_vjp(ctx.tr, x, theta, 0, theta.shape[1], u, gl, gtheta, dump, stream)
        return None, u, -gtheta


def _differentiable_inverse(tr):
    """``(y, theta) -> (x, log_det_J)`` of ``tr.inverse`` as autograd nodes on the HIP kernels: Moebius as its forward function
    on negated parameters (reference moebius.py:142-147 defines the inverse that way), the element-wise transformers through
    ``TransformerInverseFunction``, a mixed transformer with a Moebius member (reference mixed.py:165-186: plain autograd
    through every member) member by member on its own columns and parameter block, a user's torch transformer by plain
    autograd."""
    from .partial import _GatherColumns, _ReplaceColumns
    if type(tr) is MoebiusTransformer:
        return lambda y_tr, theta: TransformerFunction.apply(tr, y_tr, -theta)
    if _elementwise(tr):
        return lambda y_tr, theta: TransformerInverseFunction.apply(tr, y_tr, theta)
    if type(tr) is MixedTransformer:
        members = [_differentiable_inverse(t) for t in tr._transformers]

        def inverse(y_tr, theta):
            splits = tr.host_splits() + [theta.shape[1]]
            x, ldj = torch.zeros_like(y_tr), None
            for inv_g, ind, a, b in zip(members, tr._indices, splits[:-1], splits[1:]):
                idx = ind.to(device=y_tr.device, dtype=torch.int32)
                x_g, l_g = inv_g(_GatherColumns.apply(y_tr, idx), theta[:, a:b].contiguous())
                x = _ReplaceColumns.apply(x, x_g, idx)
                ldj = l_g if ldj is None else ldj + l_g
            return x, ldj
        return inverse
    if not isinstance(tr, _HIP_TRANSFORMERS):                # a user's torch transformer: plain autograd
        return lambda y_tr, theta: tr.inverse(y_tr, theta)
    raise NotImplementedError('tfep_amd: no backward for the inverse of ' + type(tr).__name__ +
                              ' (a subclass of a HIP-backed transformer)')


def generic_inverse(layer, y):
    """Differentiable ``AutoregressiveFlow.inverse``: the reference's algorithm (autoregressive.py:179-229: one full
    conditioner pass per autoregressive degree, the degree's columns committed after each pass, the log-det of the LAST
    pass returned) with every piece recorded by autograd -- the conditioner through its ``MaskedLinear`` modules (any other
    conditioner as the torch module it is), the transformer inverse through ``TransformerInverseFunction`` (Moebius: the
    forward function on negated parameters, as reference moebius.py:142-147 defines its inverse), the column plumbing
    through the differentiable gather / replace of ``flows/partial.py``.  This is the reference's cost (n_degrees
    conditioner passes kept for the backward), not the blocked substitution's: meant for the sizes at which the reference
    itself can train through an inverse."""
    from .partial import _GatherColumns, _ReplaceColumns
    t = layer._tables(y.device)
    tr = layer._transformer
    made = layer._conditioner

    def conditioner(x):
        cond_in = _GatherColumns.apply(x, t['cond']) if len(layer._conditioner_indices) > 0 else x
        return made.layers(made._embed(cond_in)) if isinstance(made, MADE) else made(cond_in)
    inverse = _differentiable_inverse(tr)
    if layer.has_fixed_indices:
        x = _ReplaceColumns.apply(torch.zeros_like(y), _GatherColumns.apply(y, t['fixed']), t['fixed'])
        y_tr = _GatherColumns.apply(y, t['tr'])
    else:
        x, y_tr = torch.zeros_like(y), y
    log_det_J = None
    for cols, pos in layer._inverse_steps(y.device):
        x_temp, log_det_J = inverse(y_tr, conditioner(x))
        x = _ReplaceColumns.apply(x, _GatherColumns.apply(x_temp, pos), cols)
    return x, log_det_J


class LazyInverseFunction(torch.autograd.Function):
    """``layer.inverse(y)`` under grad mode: the VALUES come from the fast path (the blocked substitution: no graph, about
    one forward of flops), and only a ``backward()`` that actually arrives pays for the differentiable route -- it re-runs
    ``generic_inverse`` under autograd from the saved ``y`` and back-propagates the incoming cotangents through it.  A
    script that samples with grad mode left on (the common case) costs what it cost under ``torch.no_grad()``; one that
    trains through the inverse gets the reference's gradients (autoregressive.py:179-229) at the reference's cost.  (First
    order only: the backward itself is not differentiable.)"""

    @staticmethod
    def forward(ctx, layer, y, *params):
        with torch.no_grad():
            x, ldj = layer._inverse_impl(y)
        ctx.layer, ctx.params = layer, params
        ctx.save_for_backward(y)
        return x, ldj

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gx, gldj):
        (y,) = ctx.saved_tensors
        layer, params = ctx.layer, ctx.params
        with torch.enable_grad():
            y_ = y.detach().requires_grad_(True)
            x2, l2 = generic_inverse(layer, y_)
            outs, cots = [], []
            if gx is not None:
                outs.append(x2)
                cots.append(gx)
            if gldj is not None:
                outs.append(l2)
                cots.append(gldj)
            wrt = [y_] + [p for p in params if p.requires_grad]
            grads = torch.autograd.grad(outs, wrt, cots, allow_unused=True)
        gy = grads[0] if ctx.needs_input_grad[1] else None
        it = iter(grads[1:])
        gparams = [next(it) if p.requires_grad else None for p in params]
        return (None, gy, *gparams)


class UnsupportedBackward(torch.autograd.Function):
    """Forward works for every configuration; differentiating an unsupported one fails loudly."""

    @staticmethod
    def forward(ctx, layer, x, *params):
        with torch.no_grad():
            return layer._forward_impl(x)

    @staticmethod
    def backward(ctx, gy, gldj):
        raise NotImplementedError(
            'tfep_amd: backward needs a transformer with a VJP kernel (affine / neural-spline / Moebius / '
            'volume-preserving / mixed) or a torch transformer that autograd can differentiate; a subclass of a HIP-backed '
            'transformer is neither.')


def _gemm(x, w, y, B, N, n_rows_w, bias=None, k_ranges=None, act=0, accumulate=0, elu_grad_of=None, tile_live=None,
          split=False, w_split=None, x_split=None, tile_list=None, tile_n=0):
    """``split``: run on split-f16 operands (x converted here with one scale per row; ``w_split`` / ``x_split`` = already
    converted ``(rows, inv_scale)`` of w / x, else w is converted here with one scale for the matrix)."""
    d = _lib.GemmDesc()
    if split:
        xs, x_inv = x_split if x_split is not None else ops.split_rows(x, x.shape[1])
        ws, w_inv = w_split if w_split is not None else ops.split_rows(w, w.shape[1], per_tensor=True)
        d.split, d.x_inv_scale, d.w_inv_scale = 1, x_inv.data_ptr(), w_inv.data_ptr()
        x, w = xs, ws
    d.x, d.ldx = x.data_ptr(), x.shape[1]
    d.w, d.ldw = w.data_ptr(), w.shape[1]
    d.bias = bias.data_ptr() if bias is not None else None
    d.k_ranges = k_ranges.data_ptr() if k_ranges is not None else None
    d.tile_order, d.col_map = None, None
    d.y, d.ldy = y.data_ptr(), y.shape[1]
    d.B, d.N, d.n_rows_w, d.k_padded, d.act, d.accumulate = B, N, n_rows_w, w.shape[1], act, accumulate
    if elu_grad_of is not None:
        d.elu_grad_of, d.ld_elu_grad_of = elu_grad_of.data_ptr(), elu_grad_of.shape[1]
    d.tile_live = tile_live.data_ptr() if tile_live is not None else None
    if tile_list is not None:
        d.tile_list, d.n_tile_list = tile_list.data_ptr(), tile_list.shape[0]
    d.tile_n = tile_n                  # (0 = the default 256 columns; the tables above count tiles of this width)
    if not split and ops.few_wide_tiles(B, N):
        # a cfg1-sized product is one or two 256 x 256 tiles: one workgroup walks the whole k range while 255 CUs idle
        # (130 us for a 224 x 224 x 1024 grad_weight).  The 32-column tile spreads it over the columns; the mask tables are
        # per 256-column tile and only save work (masked weights are zeros, masked gradients are dropped later): dense.
        d.tile_n, d.k_ranges, d.tile_live = ops.narrow_tile_n(), None, None
        d.tile_list, d.n_tile_list = None, 0
    _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), _lib.stream_of(x))
    return y


def _transpose(src, rows, cols, out):
    """out (cols_pad x rows_pad, zero filled by the caller) <- src[:rows, :cols]^T."""
    _lib.call('tfep_transpose', _lib.ptr(src), src.shape[1], rows, cols, _lib.ptr(out), out.shape[1], _lib.stream_of(src))
    return out


def _backward_plan(layer, device):
    """Host-side plan of the backward GEMMs (built once per device).

    Output-layer rows are re-packed FEATURE-MAJOR in degree-sorted feature order (row = slot*P + p) for affine
    and spline transformers: the rows that feed a tile of hidden units are then one contiguous
    range, so ``grad_input = g W`` skips the masked half of its contraction like the forward does.
    """
    key = ('bwd', str(device))
    bp = layer._dev.get(key)
    if bp is not None:
        return bp
    made = layer._conditioner
    mplan = made.plan(device)
    tm, tn, tk = ops.tile_sizes()
    lins = made._linears()
    L = len(lins) - 1
    tables = layer._tables(device)
    n_tr = tables['n_tr']
    n_out = lins[-1].out_features
    P = n_out // n_tr
    i32 = dict(device=device, dtype=torch.int32)
    bp = dict(sorted_out=type(layer._transformer) in (AffineTransformer, NeuralSplineTransformer))
    k_ranges = list(mplan['k_ranges'])
    if bp['sorted_out']:
        deg_tr = made._degrees[-1][:n_tr].cpu()
        order = torch.argsort(deg_tr, stable=True)                 # slot -> transformed feature
        slot_of = torch.empty_like(order)
        slot_of[order] = torch.arange(n_tr, device='cpu')
        p = torch.arange(P, device='cpu').repeat_interleave(n_tr)
        row_of_out = (slot_of.repeat(P) * P + p).to(**i32)         # row of reference output o = p*n_tr + t
        bp.update(order=order.to(**i32), row_of_out=row_of_out)
        n_out_pad = ops.round_up(n_out, tk)
        k_ranges[L] = ops.mask_k_ranges(lins[L].mask, tn, (n_out_pad + tn - 1) // tn, mplan['k_pad'][L], row_of_out,
                                        mplan['col_of_in'][L])
    bp['k_ranges'] = k_ranges
    use_list = os.environ.get('TFEP_TILE_LIST', '1') != '0'

    def tables(col_width):
        """Per layer: live output tiles of grad_weight (256 rows of W x ``col_width`` columns of k), the same as an
        XCD-balanced launch list, and the W-row range each column tile of grad_input needs."""
        dx_ranges, live, live_list = [], [], []
        for li, lin in enumerate(lins):
            kr = k_ranges[li].cpu().long()                            # per 256-row tile of W: [kb, ke)
            n_pad = ops.round_up(lin.out_features, tk) if li == L else mplan['n_pad'][li]
            k_pad = mplan['k_pad'][li]
            n_col_tiles = (k_pad + col_width - 1) // col_width
            lo = torch.arange(n_col_tiles) * col_width
            hi = lo + col_width
            hit = (kr[:, 0:1] < hi[None, :]) & (kr[:, 1:2] > lo[None, :])          # (row tiles, col tiles)
            live.append(hit.to(torch.uint8).contiguous().to(device))
            # the same tiles as a launch list that gives every XCD an equal share (grad_weight: all tiles cost the same)
            live_list.append(ops.xcd_balanced_tile_list(hit).to(device) if use_list else None)
            # grad_input GEMM: output column tile j (over k) needs the rows n of the tiles that touch it
            rng = torch.zeros(n_col_tiles, 2, dtype=torch.int32)
            for j in range(n_col_tiles):
                rows = torch.nonzero(hit[:, j]).flatten()
                if len(rows):
                    rng[j, 0] = int(rows.min()) * tn
                    rng[j, 1] = min((int(rows.max()) + 1) * tn, n_pad)
            dx_ranges.append(rng.to(device))
        return dict(dx_ranges=dx_ranges, live=live, live_list=live_list)

    bp.update(tables(tn))
    # The split-f16 kernel's 400-column tile for the plain linear products of a training step (the saving forward's output
    # layer, grad_input, grad_weight): the same tables counted in tiles of that width.
    bp['wide'] = None
    if os.environ.get('TFEP_SPLIT_WIDE_TILE', '1') != '0':
        tw = ops.split_wide_tile_n()
        wide = tables(tw)
        wide['tile_n'] = tw
        row_of_out = bp.get('row_of_out', mplan['row_of_out'][L])
        n_out_pad = ops.round_up(n_out, tk)
        wide['k_ranges_out'] = ops.mask_k_ranges(lins[L].mask, tw, (n_out_pad + tw - 1) // tw, mplan['k_pad'][L], row_of_out,
                                                 mplan['col_of_in'][L])
        bp['wide'] = wide
    layer._dev[key] = bp
    return bp


def _dims(layer, dev):
    made = layer._conditioner
    mplan = made.plan(dev)
    lins = made._linears()
    L = len(lins) - 1
    tm, tn, tk = ops.tile_sizes()
    n_out = lins[-1].out_features
    n_out_pad = ops.round_up(n_out, tk)
    n_pad = [mplan['n_pad'][l] for l in range(L)] + [n_out_pad]
    return mplan, lins, L, n_out, n_out_pad, n_pad, list(mplan['k_pad'])


def _weights(layer, dev):
    """Packed (degree sorted, padded) weights of every layer, their transposes, and the split-f16 versions of both, in
    the backward's row order.  Kept on the layer until a parameter or mask changes (``Tensor._version``): the
    activation-saving forward and the backward of the same step share one preparation."""
    made = layer._conditioner
    made.begin_call()            # a pack request that does not come through MADE.forward: a fresh parameter checksum for this call
    mplan, lins, L, n_out, n_out_pad, n_pad, k_pad = _dims(layer, dev)
    bplan = _backward_plan(layer, dev)
    split = layer._use_split_gemm()
    key = ('bwd_weights', str(dev))
    versions = (made._param_versions(), split)
    capturing = torch.cuda.is_current_stream_capturing()
    cached = layer._dev.get(key)
    if cached is not None and cached['versions'] == versions and not capturing:
        return cached
    f32 = dict(dtype=torch.float32, device=dev)
    W, WT, bias, Ws, WTs = [], [], [], [], []
    direct = split and os.environ.get('TFEP_DIRECT_SPLIT_PACK', '1') != '0'
    for l, lin in enumerate(lins):
        row_of_out = bplan['row_of_out'] if (l == L and bplan['sorted_out']) else None
        if direct:
            # Split-f16 operands for every GEMM of the step (the same fp32-equivalent kernel as the forward pass), straight
            # from the parameters: the weight-norm / mask / permutation pass writes split rows (one scale per matrix, from
            # max |g|), and the transpose is made from those halves (exact: a matrix and its transpose share the scale) --
            # no fp32 copy of the packed matrix, no maximum pass, no conversion pass: 8.2 -> 5.1 ms per re-pack of a cfg2
            # output layer
            ws, w_inv, b, _ = made._pack_layer_split(mplan, l, lin, row_of_out=row_of_out, n_rows=n_pad[l])
            wts_l = torch.empty(k_pad[l], n_pad[l], **f32)
            _lib.call('tfep_transpose_split_rows', _lib.ptr(ws), ws.shape[1], n_pad[l], k_pad[l], _lib.ptr(wts_l), n_pad[l],
                      _lib.stream_of(ws))
            W.append(ws)                      # (the GEMM launches read shapes from W; the operands are Ws / WTs)
            bias.append(b)
            WT.append(None)
            Ws.append((ws, w_inv))
            WTs.append((wts_l, w_inv))
            continue
        w, b = made._pack_layer(mplan, l, lin, row_of_out=row_of_out, n_rows=n_pad[l])
        W.append(w)
        bias.append(b)
        if not split:
            WT.append(_transpose(w, n_pad[l], k_pad[l], ops.zeros(k_pad[l], n_pad[l], **f32)))
            Ws.append(None)
            WTs.append(None)
            continue
        # (TFEP_DIRECT_SPLIT_PACK=0: fp32 pack, then split rows and their transpose in one pass each)
        WT.append(None)
        Ws.append(ops.split_rows(w, w.shape[1], per_tensor=True))
        wts_l = torch.empty(k_pad[l], n_pad[l], **f32)
        _lib.call('tfep_transpose_split', _lib.ptr(w), w.shape[1], n_pad[l], k_pad[l], _lib.ptr(wts_l), n_pad[l],
                  n_pad[l], 0, _lib.ptr(Ws[l][1]), None, _lib.stream_of(w))
        WTs.append((wts_l, Ws[l][1]))
    wts = dict(versions=versions, split=split, W=W, WT=WT, bias=bias, Ws=Ws, WTs=WTs)
    if _SAVE_BYTES > 0 and not capturing:
        layer._dev[key] = wts
    return wts


def _conditioner_forward(layer, wts, cin, Bc):
    """Hidden activations (``h[0]`` = the zero-padded conditioner input) and transformer parameters, backward row order."""
    dev = cin.device
    mplan, lins, L, n_out, n_out_pad, n_pad, k_pad = _dims(layer, dev)
    bplan = _backward_plan(layer, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    split = wts['split']
    h = [ops.pad_columns(cin, k_pad[0])]
    for l in range(L):
        h.append(_gemm(h[-1], wts['W'][l], torch.empty(Bc, n_pad[l], **f32), Bc, n_pad[l], n_pad[l], bias=wts['bias'][l],
                       k_ranges=mplan['k_ranges'][l], act=1, split=split, w_split=wts['Ws'][l]))
    theta = torch.empty(Bc, n_out_pad, **f32)
    wide = bplan['wide'] if split else None
    _gemm(h[-1], wts['W'][L], theta, Bc, n_out_pad, n_pad[L], bias=wts['bias'][L],
          k_ranges=wide['k_ranges_out'] if wide else bplan['k_ranges'][L], tile_n=wide['tile_n'] if wide else 0,
          split=split, w_split=wts['Ws'][L])
    return h, theta


def saves_activations(layer, x):
    """Whether the training forward of ``layer`` keeps its activations for the backward (see ``_SAVE_BYTES``)."""
    return x.dim() == 2 and saves_activations_at(layer, x.shape[0])


#: set by ``tfep_amd.graphs.GraphedTrainingStep`` around its warm-up and capture: inside a capture the layers recompute their
#: activations, and the warm-up has to build the plans of exactly that path
FORCE_RECOMPUTE = False


def saves_activations_at(layer, batch):
    if _SAVE_BYTES <= 0 or FORCE_RECOMPUTE or not supported(layer) or torch.cuda.is_current_stream_capturing():
        return False
    if type(layer._transformer) not in (AffineTransformer, NeuralSplineTransformer):
        return False
    emb = getattr(layer._conditioner, 'embedding', None)
    if emb is not None and type(emb) is not PeriodicEmbedding:
        return False
    lins = layer._conditioner._linears()
    need = int(batch) * 4 * (lins[-1].out_features + sum(lin.out_features for lin in lins[:-1]))
    if not (1 << 20) <= int(batch) * lins[-1].out_features * 4 <= _SAVE_BYTES:
        return False                 # (below a MiB of parameters the step is launch bound: the fused forward + recompute wins)
    # ... and only a modest share of what the device still has (a deep flow keeps this for every layer until its backward
    # has run; the weight packings and gradient buffers of the backward need room too): otherwise recompute
    dev = lins[-1].bias.device
    if dev.type != 'cuda':
        return False
    free = torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)
    return need <= 0.15 * free


def _fused_saving_plan(layer, dev):
    """Tables of the fused output-layer launch on the BACKWARD's packing (feature-major rows in degree-sorted slot order),
    or None when the layer's forward has no fused split kernel of that form (anything but a plain RQ-spline transformer
    in one of the fused layouts)."""
    key = ('fused_saving', str(dev))
    if key in layer._dev:
        return layer._dev[key]
    from ..transformers.spline import NeuralSplineTransformer
    from .autoregressive import _FUSED_SPLINE
    plan = None
    bplan = _backward_plan(layer, dev)
    if type(layer._transformer) is NeuralSplineTransformer and bplan['sorted_out'] and layer._fused_kind() == _FUSED_SPLINE \
            and os.environ.get('TFEP_FUSED_SAVING', '1') != '0':
        made = layer._conditioner
        mplan, lins, L, n_out, n_out_pad, n_pad, k_pad = _dims(layer, dev)
        grp = layer._fused_plan(dev, _FUSED_SPLINE, layer._tables(dev))['groups'][0]
        P = grp['P']
        n_tr = layer._tables(dev)['n_tr']
        # the fused plan and the backward sort the features the same way (stable argsort of their degrees)
        same = torch.equal(grp['feat_tr'][:n_tr].cpu().long(), bplan['order'].cpu().long())
        desc = layer._transformer.config(dev).desc
        if same and P * n_tr == n_out and _lib.load().tfep_fused_saving_supported(ctypes.byref(desc)):
            n_tiles = grp['n_slots'] // 16
            k_ranges = ops.mask_k_ranges(lins[L].mask, 16 * P, n_tiles, k_pad[L], bplan['row_of_out'], mplan['col_of_in'][L])
            plan = dict(grp=grp, P=P, k_ranges=k_ranges, tile_order=ops.heavy_first_order(k_ranges))
    layer._dev[key] = plan
    return plan


def forward_saving(layer, x):
    """``(y, log_det_J, saved)`` of the layer on the backward's weight packing; ``saved`` holds the hidden activations and
    the transformer parameters for ``layer_backward``.  RQ splines on split-f16 operands: the output layer, the spline and
    the parameter store are ONE launch of the fused kernel (``tfep_fused_output_transformer_forward_split_saving``);
    otherwise the un-fused kernels."""
    x, ldx = _lib.rows(x, 'x')
    dev = x.device
    B, D = x.shape
    made = layer._conditioner
    emb = getattr(made, 'embedding', None)
    tables = layer._tables(dev)
    bplan = _backward_plan(layer, dev)
    mplan, lins, L, n_out, n_out_pad, n_pad, k_pad = _dims(layer, dev)
    n_tr = tables['n_tr']
    P = n_out // n_tr
    wts = _weights(layer, dev)
    cin = emb(x) if emb is not None else x
    fs = _fused_saving_plan(layer, dev) if wts['split'] else None
    if fs is not None:
        f32 = dict(dtype=torch.float32, device=dev)
        h = [ops.pad_columns(cin, k_pad[0])]
        for l in range(L):
            h.append(_gemm(h[-1], wts['W'][l], torch.empty(B, n_pad[l], **f32), B, n_pad[l], n_pad[l], bias=wts['bias'][l],
                           k_ranges=mplan['k_ranges'][l], act=1, split=True, w_split=wts['Ws'][l]))
        hs, h_inv = ops.split_rows(h[-1], h[-1].shape[1])
        theta = torch.empty(B, n_out_pad, **f32)
        if n_out_pad > n_out:
            theta[:, n_out:] = 0.0
        y = x.clone() if layer.has_fixed_indices else torch.empty(B, D, **f32)
        ldj = torch.empty(B, **f32)
        grp = fs['grp']
        ws = torch.empty(grp['n_slots'] // 16, B, dtype=torch.float64, device=dev)
        desc = layer._transformer.config(dev).desc
        w_s, w_inv = wts['Ws'][L]
        _lib.call('tfep_fused_output_transformer_forward_split_saving', _lib.ptr(hs), hs.shape[1], _lib.ptr(h_inv),
                  _lib.ptr(w_s), w_s.shape[1], _lib.ptr(w_inv), _lib.ptr(wts['bias'][L]), _lib.ptr(fs['k_ranges']),
                  _lib.ptr(fs['tile_order']), ctypes.byref(desc), _lib.ptr(x), ldx, _lib.ptr(y), D,
                  _lib.ptr(grp['feat_index']), _lib.ptr(grp['feat_tr']), grp['n_slots'], _lib.ptr(ws), _lib.ptr(ldj), 0, B,
                  n_pad[L], k_pad[L], 1, _lib.ptr(theta), n_out_pad, _lib.stream_of(x))
        return y, ldj, dict(h=h, theta=theta)
    h, theta = _conditioner_forward(layer, wts, cin, B)
    x_tr = ops.gather_columns(x, tables['tr']) if layer.has_fixed_indices else x
    xs = ops.gather_columns(x_tr, bplan['order'])
    ys, ldj = transformer_forward(layer._transformer, xs, theta, (n_out_pad, 1, P), bplan['order'], _lib.stream_of(x))
    if layer.has_fixed_indices:
        y_tr = torch.empty(B, n_tr, dtype=torch.float32, device=dev)
        ops.scatter_columns(ys, bplan['order'], y_tr)
        y = x.clone()
        ops.scatter_columns(y_tr, tables['tr'], y)
    else:
        y = torch.empty(B, D, dtype=torch.float32, device=dev)
        ops.scatter_columns(ys, bplan['order'], y)
    return y, ldj, dict(h=h, theta=theta)


def layer_backward(layer, x, gy, gldj, saved=None, need_gx=True):
    """Returns (gx, [grads in trainable_tensors() order]).  ``saved``: activations kept by ``forward_saving``.
    ``need_gx`` False: the gradient w.r.t. the layer's input is not wanted (None is returned for it)."""
    if not supported(layer):
        raise NotImplementedError(
            'tfep_amd: backward needs a transformer with a VJP kernel (affine / neural-spline / Moebius / '
            'volume-preserving / mixed) or a torch transformer that autograd can differentiate; a subclass of a HIP-backed '
            'transformer is neither.')
    x, _ = _lib.rows(x, 'x')
    gy = gy.contiguous().float()
    gldj = gldj.contiguous().float() if gldj is not None else None
    dev = x.device
    B, D = x.shape
    made = layer._conditioner
    emb = getattr(made, 'embedding', None)
    tr = layer._transformer
    tables = layer._tables(dev)
    bplan = _backward_plan(layer, dev)
    mplan, lins, L, n_out, n_out_pad, n_pad, k_pad = _dims(layer, dev)
    tm, tn, tk = ops.tile_sizes()
    f32 = dict(dtype=torch.float32, device=dev)
    n_tr = tables['n_tr']
    P = n_out // n_tr
    stream = _lib.stream_of(x)

    # ---- weights: packed (degree sorted, padded) and their transposes (shared with forward_saving of the same step)
    wts = _weights(layer, dev)
    W, WT, bias, Ws, WTs, split = wts['W'], wts['WT'], wts['bias'], wts['Ws'], wts['WTs'], wts['split']
    sorted_out = bplan['sorted_out']
    # (ops.zeros: fill kernels, not memsets -- a captured training step replays them; see ops.zeros)
    # The packed weight gradients are written (first chunk) or accumulated (later chunks) over the live tiles only, and only
    # those are read afterwards (weight_norm_backward*: the live prefix, or a select on the mask): no 5.5 GB clear.
    gW = [(ops.zeros if B == 0 else torch.empty)(n_pad[l], k_pad[l], **f32) for l in range(L + 1)]      # (empty batch: no chunk writes)
    gb = [ops.zeros(n_pad[l], **f32) for l in range(L + 1)]
    gx = torch.empty(B, D, **f32) if need_gx else None
    emb_generic = emb is not None and type(emb) is not PeriodicEmbedding
    emb_params = _embedding_params(layer)
    g_emb = [torch.zeros_like(p) for p in emb_params]

    chunk = max(tm, min(B, (_CHUNK_BYTES // (4 * n_out_pad)) // tm * tm))
    for b0 in range(0, B, chunk):
        b1 = min(B, b0 + chunk)
        Bc = b1 - b0
        Bc_pad = ops.round_up(Bc, tk)
        xc, gyc = x[b0:b1], gy[b0:b1]
        glc = gldj[b0:b1] if gldj is not None else None

        # ---- recompute the conditioner forward for the chunk
        if emb_generic:
            # any other MAFEmbedding (flip-invariant / mixed): differentiated by autograd, chunk by chunk
            with torch.enable_grad():
                x_emb = xc.detach().requires_grad_(True)
                cin_graph = emb(x_emb)
            cin = cin_graph.detach()
        else:
            cin = emb(xc) if emb is not None and saved is None else xc
        if saved is not None:                                  # kept by the forward of this step
            h, theta = [t[b0:b1] for t in saved['h']], saved['theta'][b0:b1]
        else:
            h, theta = _conditioner_forward(layer, wts, cin, Bc)

        # ---- transformer VJP: gtheta (reference parameter layout, zero padded columns), direct gx
        if layer.has_fixed_indices:
            x_tr, gy_tr = ops.gather_columns(xc, tables['tr']), ops.gather_columns(gyc, tables['tr'])
        else:
            x_tr, gy_tr = xc, gyc
        if sorted_out:
            # affine / spline: the kernel writes every parameter of every feature; only the padding columns need zeros
            gtheta = torch.empty(Bc, n_out_pad, **f32)
            if n_out_pad > n_out:
                gtheta[:, n_out:] = 0.0
        else:
            gtheta = ops.zeros(Bc, n_out_pad, **f32)
        gx_dir = torch.empty(Bc, n_tr, **f32)
        x_tr, gy_tr = x_tr.contiguous(), gy_tr.contiguous()
        if sorted_out:
            # theta / gtheta are feature-major in slot order: work on slot-ordered copies of x and gy
            xs, gys = ops.gather_columns(x_tr, bplan['order']), ops.gather_columns(gy_tr, bplan['order'])
            gxs = torch.empty(Bc, n_tr, **f32)
            transformer_vjp(tr, xs, theta, 0, n_out_pad, gys, glc, gtheta, gxs, stream, layout=(n_out_pad, 1, P),
                            order=bplan['order'])
            ops.scatter_columns(gxs, bplan['order'], gx_dir)
        else:
            transformer_vjp(tr, x_tr, theta, 0, n_out_pad, gy_tr, glc, gtheta, gx_dir, stream)
        del theta

        # ---- masked linears, last to first.  g = gradient w.r.t. the layer's pre-activation output.
        g = gtheta
        for l in range(L, -1, -1):
            if split:
                # bias gradient and the row scales of g^T from one pass over g; g^T and h^T straight to split rows (no fp32
                # transposes: the 4.9 GB gradient of the transformer parameters went through HBM nine times before)
                cmax = torch.empty(n_pad[l], **f32)
                _lib.call('tfep_column_sums_absmax', _lib.ptr(g), g.shape[1], Bc, n_pad[l], _lib.ptr(gb[l]), 1, _lib.ptr(cmax), stream)
                gTs, gT_inv = torch.empty(n_pad[l], Bc_pad, **f32), torch.empty(n_pad[l], **f32)
                _lib.call('tfep_transpose_split', _lib.ptr(g), g.shape[1], Bc, n_pad[l], _lib.ptr(gTs), Bc_pad, Bc_pad, 2,
                          _lib.ptr(cmax), _lib.ptr(gT_inv), stream)
                hTs, hT_inv = torch.empty(k_pad[l], Bc_pad, **f32), torch.empty(2, **f32)
                _lib.call('tfep_transpose_split', _lib.ptr(h[l]), h[l].shape[1], Bc, k_pad[l], _lib.ptr(hTs), Bc_pad, Bc_pad, 1,
                          None, _lib.ptr(hT_inv), stream)
                # grad_weight (packed) += g^T h   [rows n, cols k], masked tiles skipped
                tbl = bplan['wide'] or bplan
                _gemm(gTs, hTs, gW[l], n_pad[l], k_pad[l], k_pad[l], accumulate=int(b0 > 0), tile_live=tbl['live'][l], split=True,
                      x_split=(gTs, gT_inv), w_split=(hTs, hT_inv), tile_list=tbl['live_list'][l], tile_n=tbl.get('tile_n', 0))
                del gTs, hTs, cmax
            else:
                _lib.call('tfep_column_sums', _lib.ptr(g), g.shape[1], Bc, n_pad[l], _lib.ptr(gb[l]), 1, stream)
                gT = _transpose(g, Bc, n_pad[l], ops.zeros(n_pad[l], Bc_pad, **f32))
                hT = _transpose(h[l], Bc, k_pad[l], ops.zeros(k_pad[l], Bc_pad, **f32))
                _gemm(gT, hT, gW[l], n_pad[l], k_pad[l], k_pad[l], accumulate=int(b0 > 0), tile_live=bplan['live'][l],
                      tile_list=bplan['live_list'][l])
                del gT, hT
            # grad_input = g W  (x ELU'(h) for hidden inputs)
            if l == 0 and not need_gx and not emb_params:
                break                                   # nothing upstream wants the gradient of the conditioner's input
            gin = torch.empty(Bc, k_pad[l], **f32)
            tbl = (bplan['wide'] if split else None) or bplan
            _gemm(g, WT[l], gin, Bc, k_pad[l], k_pad[l], k_ranges=tbl['dx_ranges'][l],
                  elu_grad_of=h[l] if l > 0 else None, split=split, w_split=WTs[l], tile_n=tbl.get('tile_n', 0))
            g = gin

        # ---- gradient w.r.t. the layer input: through the conditioner + direct
        if not need_gx and not emb_params:
            continue
        if emb_generic:
            g_in = torch.autograd.grad(cin_graph, [x_emb] + emb_params, g[:, :cin.shape[1]].contiguous(), allow_unused=True)
            gxc = g_in[0].contiguous() if g_in[0] is not None else torch.zeros(Bc, D, **f32)
            for acc_g, new_g in zip(g_emb, g_in[1:]):
                if new_g is not None:
                    acc_g += new_g
            del cin_graph, x_emb
        elif emb is not None:
            per, non = emb.device_indices(dev)
            gxc = ops.zeros(Bc, D, **f32)
            _lib.call('tfep_periodic_embedding_backward', _lib.ptr(xc), xc.shape[1] if Bc > 1 else D, _lib.ptr(per),
                      per.numel(), _lib.ptr(non), non.numel(), *emb.host_limits(),
                      _lib.ptr(g), g.shape[1], _lib.ptr(gxc), D, Bc, stream)
        else:
            gxc = g[:, :D].contiguous()
        if layer.has_fixed_indices:
            direct = gyc.clone()                                  # fixed features: y = x
            ops.scatter_columns(gx_dir, tables['tr'], direct)
        else:
            direct = gx_dir
        if need_gx:
            _lib.call('tfep_add_inplace', _lib.ptr(direct), direct.shape[1], _lib.ptr(gxc), D, Bc, D, stream)
            gx[b0:b1] = gxc

    # ---- packed weight / bias gradients -> parameter gradients
    grads = []
    for l, lin in enumerate(lins):
        row_of_out = bplan['row_of_out'] if (l == L and sorted_out) else mplan['row_of_out'][l]
        col_of_in = mplan['col_of_in'][l]
        # long rows under a prefix mask: the LDS-staged kernel (mask not read, packed gradient read over the live prefix)
        cut = made._mask_prefix_cuts(mplan, l, lin) if 8192 <= lin.in_features <= 16384 and \
            os.environ.get('TFEP_PACK_LDS', '1') != '0' else None
        in_of_col = mplan['in_of_col'][l]
        prefix = cut is not None and (in_of_col is not None or col_of_in is None)
        if lin.has_weight_norm:
            gv = torch.empty_like(lin.weight_v)
            gg = torch.empty(lin.out_features, 1, **f32)
            if prefix:
                _lib.call('tfep_weight_norm_backward_prefix', _lib.ptr(gW[l]), k_pad[l], _lib.ptr(lin.weight_v.detach()),
                          _lib.ptr(lin.weight_g.detach()), lin.out_features, lin.in_features, _lib.ptr(row_of_out),
                          _lib.ptr(in_of_col), _lib.ptr(cut), _lib.ptr(gv), _lib.ptr(gg), stream)
            else:
                _lib.call('tfep_weight_norm_backward', _lib.ptr(gW[l]), k_pad[l], _lib.ptr(lin.weight_v.detach()),
                          _lib.ptr(lin.weight_g.detach()), _lib.ptr(lin.mask), lin.out_features, lin.in_features,
                          _lib.ptr(row_of_out), _lib.ptr(col_of_in), _lib.ptr(gv), _lib.ptr(gg), stream)
            grads += [gg, gv]
            # (already masked by the kernel: the parameters' gradient hooks of masked_weight_norm need not do it again)
            # The tag names THIS gradient (its storage address): a stale tag -- a backward that did not hand the
            # parameter its gradient, an exception between layers -- cannot wave a later, unrelated gradient through.
            for prm, grad in ((lin.weight_g, gg), (lin.weight_v, gv)):
                if prm.requires_grad:
                    prm.__dict__[GRAD_IS_MASKED] = grad.data_ptr()
        else:
            gw = torch.empty_like(lin._parameters['weight'])
            if prefix:
                _lib.call('tfep_weight_norm_backward_prefix', _lib.ptr(gW[l]), k_pad[l], _lib.ptr(lin._parameters['weight'].detach()),
                          None, lin.out_features, lin.in_features, _lib.ptr(row_of_out), _lib.ptr(in_of_col), _lib.ptr(cut),
                          _lib.ptr(gw), None, stream)
            else:
                _lib.call('tfep_weight_norm_backward', _lib.ptr(gW[l]), k_pad[l], _lib.ptr(lin._parameters['weight'].detach()),
                          None, _lib.ptr(lin.mask), lin.out_features, lin.in_features, _lib.ptr(row_of_out),
                          _lib.ptr(col_of_in), _lib.ptr(gw), None, stream)
            grads.append(gw)
        if row_of_out is None:
            grads.append(gb[l][:lin.out_features].clone())
        else:
            grads.append(ops.gather_columns(gb[l][None, :], row_of_out)[0])
    return gx, grads + g_emb
