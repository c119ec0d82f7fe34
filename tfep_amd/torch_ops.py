"""The kernels as custom torch ops: ``torch.ops.tfep.*`` (north star: "exposed as custom torch ops").

Each op is a thin dispatcher entry over one C-ABI entry point of ``include/tfep_hip.h`` (through ``tfep_amd.ops``):
a schema, a HIP ("cuda" device type) implementation, a fake / meta implementation (shapes and dtypes only, so that
``torch.compile``, ``make_fx`` and fake-tensor tracing see through them) and, where a VJP kernel exists, an autograd
formula whose backward is itself a registered op.  There is no CPU implementation: calling an op with CPU tensors fails
in the dispatcher ("no kernel for CPU") -- the same no-fallback policy as the rest of the package.

  tfep::affine_forward / affine_inverse / affine_backward            reference transformers/affine.py:281-363
  tfep::spline_forward / spline_inverse / spline_backward            reference transformers/spline.py:184-261, 424-564
  tfep::moebius_forward / moebius_inverse / moebius_backward         reference transformers/moebius.py:104-147, 374-478
  tfep::masked_linear / masked_linear_backward                       reference masked.py:220-302, 351-404
  tfep::fused_output_transformer                                     masked.py:188-208 (last layer) + the transformer
  tfep::tfep_reduce                                                  loss.py:125-140, analysis/estimator.py:73-86

The Module API (``tfep_amd.nn``, ``tfep_amd.loss``) routes through these ops.
"""
import ctypes
from typing import Optional, Tuple

import torch
from torch import Tensor
from torch.library import custom_op

from . import _lib, ops

_DEV = 'cuda'          # PyTorch-ROCm's name for the HIP device type


def _pair_like(x):
    return x.new_empty(x.shape), x.new_empty((x.shape[0],))


# ============================================================================= affine

@custom_op('tfep::affine_forward', mutates_args=(), device_types=_DEV)
def affine_forward(x: Tensor, parameters: Tensor) -> Tuple[Tensor, Tensor]:
    return ops.affine(x, parameters, inverse=False)


@custom_op('tfep::affine_inverse', mutates_args=(), device_types=_DEV)
def affine_inverse(y: Tensor, parameters: Tensor) -> Tuple[Tensor, Tensor]:
    return ops.affine(y, parameters, inverse=True)


@custom_op('tfep::affine_backward', mutates_args=(), device_types=_DEV)
def affine_backward(x: Tensor, parameters: Tensor, grad_y: Tensor, grad_log_det_J: Tensor) -> Tuple[Tensor, Tensor]:
    # (contiguous copies are held in locals until the launch is queued: a temporary freed inside the argument list
    # hands its block back to the caching allocator, and the NEXT temporary may be written over it)
    x, parameters, gy, gl = x.contiguous(), parameters.contiguous(), grad_y.contiguous(), grad_log_det_J.contiguous()
    B, D = x.shape
    gx = torch.empty_like(x)
    gp = ops.zeros(*parameters.shape, dtype=torch.float32, device=x.device)
    lay = _lib.ParamLayout(parameters.shape[1], D, 1)
    _lib.call('tfep_affine_backward', _lib.ptr(x), D, _lib.ptr(parameters), lay, _lib.ptr(gy), D,
              _lib.ptr(gl), _lib.ptr(gp), lay, _lib.ptr(gx), D, B, D, _lib.stream_of(x))
    return gx, gp


affine_forward.register_fake(lambda x, parameters: _pair_like(x))
affine_inverse.register_fake(lambda y, parameters: _pair_like(y))
affine_backward.register_fake(lambda x, parameters, gy, gl: (x.new_empty(x.shape), parameters.new_empty(parameters.shape)))


def _vjp_inputs(ctx, grads):
    """Upstream gradients of (y, log_det_J), zeros where autograd passes None."""
    x = ctx.saved_tensors[0]
    gy, gl = grads
    gy = torch.zeros_like(x) if gy is None else gy
    gl = x.new_zeros(x.shape[0]) if gl is None else gl
    return gy, gl


def _save_xp(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])


def _affine_bwd(ctx, gy, gl):
    x, p = ctx.saved_tensors
    gy, gl = _vjp_inputs(ctx, (gy, gl))
    return torch.ops.tfep.affine_backward(x, p, gy, gl)


affine_forward.register_autograd(_affine_bwd, setup_context=_save_xp)


# ============================================================================= rational-quadratic spline

def _spline_cfg(x0, xf, y0, yf, n_bins, circular, identity, learn_lower, learn_upper, min_bin, min_slope):
    return ops.SplineConfig(x0, xf, y0, yf, n_bins, circular, identity, learn_lower, learn_upper, min_bin, min_slope)


@custom_op('tfep::spline_forward', mutates_args=(), device_types=_DEV)
def spline_forward(x: Tensor, parameters: Tensor, x0: Tensor, xf: Tensor, y0: Tensor, yf: Tensor, n_bins: int,
                   circular: bool, identity_boundary_slopes: bool, learn_lower_bound: bool, learn_upper_bound: bool,
                   min_bin_size: float, min_slope: float) -> Tuple[Tensor, Tensor]:
    cfg = _spline_cfg(x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound, learn_upper_bound,
                      min_bin_size, min_slope)
    return ops.spline(x, parameters, cfg, inverse=False)


@custom_op('tfep::spline_inverse', mutates_args=(), device_types=_DEV)
def spline_inverse(y: Tensor, parameters: Tensor, x0: Tensor, xf: Tensor, y0: Tensor, yf: Tensor, n_bins: int,
                   circular: bool, identity_boundary_slopes: bool, learn_lower_bound: bool, learn_upper_bound: bool,
                   min_bin_size: float, min_slope: float) -> Tuple[Tensor, Tensor]:
    cfg = _spline_cfg(x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound, learn_upper_bound,
                      min_bin_size, min_slope)
    return ops.spline(y, parameters, cfg, inverse=True)


@custom_op('tfep::spline_backward', mutates_args=(), device_types=_DEV)
def spline_backward(x: Tensor, parameters: Tensor, grad_y: Tensor, grad_log_det_J: Tensor, x0: Tensor, xf: Tensor,
                    y0: Tensor, yf: Tensor, n_bins: int, circular: bool, identity_boundary_slopes: bool,
                    learn_lower_bound: bool, learn_upper_bound: bool, min_bin_size: float,
                    min_slope: float) -> Tuple[Tensor, Tensor]:
    cfg = _spline_cfg(x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound, learn_upper_bound,
                      min_bin_size, min_slope)
    x, parameters, gy, gl = x.contiguous(), parameters.contiguous(), grad_y.contiguous(), grad_log_det_J.contiguous()
    B, D = x.shape
    gx = torch.empty_like(x)
    gp = ops.zeros(*parameters.shape, dtype=torch.float32, device=x.device)
    lay = _lib.ParamLayout(parameters.shape[1], D, 1)
    _lib.call('tfep_spline_backward', _lib.ptr(x), D, _lib.ptr(parameters), lay, ctypes.byref(cfg.desc),
              _lib.ptr(gy), D, _lib.ptr(gl), _lib.ptr(gp), lay, _lib.ptr(gx), D, B, D, _lib.stream_of(x))
    return gx, gp


spline_forward.register_fake(lambda x, parameters, *cfg: _pair_like(x))
spline_inverse.register_fake(lambda y, parameters, *cfg: _pair_like(y))
spline_backward.register_fake(lambda x, parameters, gy, gl, *cfg: (x.new_empty(x.shape),
                                                                   parameters.new_empty(parameters.shape)))


def _spline_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1], *inputs[2:6])
    ctx.cfg = tuple(inputs[6:])


def _spline_bwd(ctx, gy, gl):
    x, p, x0, xf, y0, yf = ctx.saved_tensors
    gy, gl = _vjp_inputs(ctx, (gy, gl))
    gx, gp = torch.ops.tfep.spline_backward(x, p, gy, gl, x0, xf, y0, yf, *ctx.cfg)
    return (gx, gp) + (None,) * 11


spline_forward.register_autograd(_spline_bwd, setup_context=_spline_setup)


# ============================================================================= Moebius

@custom_op('tfep::moebius_forward', mutates_args=(), device_types=_DEV)
def moebius_forward(x: Tensor, parameters: Tensor, dimension: int, max_radius: float,
                    unit_sphere: bool) -> Tuple[Tensor, Tensor]:
    return ops.moebius(x, parameters, dimension, max_radius, unit_sphere, inverse=False)


@custom_op('tfep::moebius_inverse', mutates_args=(), device_types=_DEV)
def moebius_inverse(y: Tensor, parameters: Tensor, dimension: int, max_radius: float,
                    unit_sphere: bool) -> Tuple[Tensor, Tensor]:
    return ops.moebius(y, parameters, dimension, max_radius, unit_sphere, inverse=True)


@custom_op('tfep::moebius_backward', mutates_args=(), device_types=_DEV)
def moebius_backward(x: Tensor, parameters: Tensor, grad_y: Tensor, grad_log_det_J: Tensor, dimension: int,
                     max_radius: float, unit_sphere: bool) -> Tuple[Tensor, Tensor]:
    x, parameters, gy, gl = x.contiguous(), parameters.contiguous(), grad_y.contiguous(), grad_log_det_J.contiguous()
    B, D = x.shape
    gx = torch.empty_like(x)
    gp = ops.zeros(*parameters.shape, dtype=torch.float32, device=x.device)
    _lib.call('tfep_moebius_backward', _lib.ptr(x), D, _lib.ptr(parameters), parameters.shape[1], int(dimension),
              float(max_radius), int(bool(unit_sphere)), 1, _lib.ptr(gy), D, _lib.ptr(gl), _lib.ptr(gp),
              parameters.shape[1], _lib.ptr(gx), D, B, D, _lib.stream_of(x))
    return gx, gp


moebius_forward.register_fake(lambda x, parameters, *a: _pair_like(x))
moebius_inverse.register_fake(lambda y, parameters, *a: _pair_like(y))
moebius_backward.register_fake(lambda x, parameters, gy, gl, *a: (x.new_empty(x.shape),
                                                                  parameters.new_empty(parameters.shape)))


def _moebius_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[0], inputs[1])
    ctx.cfg = tuple(inputs[2:])


def _moebius_bwd(ctx, gy, gl):
    x, p = ctx.saved_tensors
    gy, gl = _vjp_inputs(ctx, (gy, gl))
    gx, gp = torch.ops.tfep.moebius_backward(x, p, gy, gl, *ctx.cfg)
    return gx, gp, None, None, None


moebius_forward.register_autograd(_moebius_bwd, setup_context=_moebius_setup)


# ============================================================================= masked linear

@custom_op('tfep::masked_linear', mutates_args=(), device_types=_DEV)
def masked_linear(input: Tensor, weight: Tensor, bias: Optional[Tensor], mask: Optional[Tensor],
                  weight_g: Optional[Tensor]) -> Tensor:
    x2 = input.reshape(-1, input.shape[-1])
    n_out, k = weight.shape
    # (F.linear's shape errors: a narrower input would otherwise be zero padded up to the tile size without a word)
    if input.shape[-1] != k:
        raise RuntimeError(f'masked_linear: input has {input.shape[-1]} features, weight is {n_out} x {k}')
    if mask is not None and tuple(mask.shape) != (n_out, k):
        raise RuntimeError(f'masked_linear: mask is {tuple(mask.shape)}, weight is {n_out} x {k}')
    if bias is not None and bias.numel() != n_out:
        raise RuntimeError(f'masked_linear: bias has {bias.numel()} entries for {n_out} output features')
    if weight_g is not None and weight_g.numel() != n_out:
        raise RuntimeError(f'masked_linear: weight_g has {weight_g.numel()} entries for {n_out} output features')
    tm, tn, tk = ops.tile_sizes()
    k_pad, n_pad = ops.round_up(k, tk), ops.round_up(n_out, tk)
    w = ops.masked_weight_prepare(weight, weight_g, mask, n_rows_padded=n_pad, k_padded=k_pad)
    y = ops.masked_linear_packed(ops.pad_columns(x2, k_pad), w, bias, n_out)
    return y.reshape(*input.shape[:-1], n_out)


@masked_linear.register_fake
def _(input, weight, bias, mask, weight_g):
    return input.new_empty((*input.shape[:-1], weight.shape[0]))


@custom_op('tfep::masked_linear_backward', mutates_args=(), device_types=_DEV)
def masked_linear_backward(grad_output: Tensor, input: Tensor, weight: Tensor, mask: Optional[Tensor],
                           weight_g: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """``(grad_input, grad_weight, grad_bias, grad_weight_g)``; grad_weight_g is empty (0 elements) without weight norm."""
    from .nn.flows._backward import _gemm, _transpose
    n_out, k = weight.shape
    tm, tn, tk = ops.tile_sizes()
    k_pad, n_pad = ops.round_up(k, tk), ops.round_up(n_out, tk)
    f32 = dict(dtype=torch.float32, device=input.device)
    w = ops.masked_weight_prepare(weight, weight_g, mask, n_rows_padded=n_pad, k_padded=k_pad)
    xp = ops.pad_columns(input.reshape(-1, k), k_pad)
    g2 = grad_output.reshape(-1, n_out).float()
    B = g2.shape[0]
    gp = ops.pad_columns(g2, n_pad)
    stream = _lib.stream_of(xp)
    wt = _transpose(w, n_pad, k_pad, torch.zeros(k_pad, n_pad, **f32))
    gx = _gemm(gp, wt, torch.empty(B, k_pad, **f32), B, k_pad, k_pad)
    grad_input = gx[:, :k].reshape(input.shape).contiguous()
    Bp = ops.round_up(B, tk)
    gT = _transpose(gp, B, n_pad, torch.zeros(n_pad, Bp, **f32))
    xT = _transpose(xp, B, k_pad, torch.zeros(k_pad, Bp, **f32))
    gw = _gemm(gT, xT, torch.zeros(n_pad, k_pad, **f32), n_pad, k_pad, k_pad, accumulate=1)
    grad_weight = torch.empty(n_out, k, **f32)
    grad_g = torch.empty(n_out, 1, **f32) if weight_g is not None else torch.empty(0, **f32)
    wc = weight.contiguous()
    gc = None if weight_g is None else weight_g.contiguous()
    mc = None if mask is None else mask.contiguous()
    _lib.call('tfep_weight_norm_backward', _lib.ptr(gw), k_pad, _lib.ptr(wc), _lib.ptr(gc), _lib.ptr(mc), n_out, k, None,
              None, _lib.ptr(grad_weight), _lib.ptr(grad_g) if weight_g is not None else None, stream)
    grad_bias = torch.empty(n_out, **f32)
    _lib.call('tfep_column_sums', _lib.ptr(gp), n_pad, B, n_out, _lib.ptr(grad_bias), 0, stream)
    return grad_input, grad_weight, grad_bias, grad_g


@masked_linear_backward.register_fake
def _(grad_output, input, weight, mask, weight_g):
    n_out = weight.shape[0]
    gg = weight.new_empty((n_out, 1)) if weight_g is not None else weight.new_empty((0,))
    return input.new_empty(input.shape), weight.new_empty(weight.shape), weight.new_empty((n_out,)), gg


def _ml_setup(ctx, inputs, output):
    input, weight, bias, mask, weight_g = inputs
    ctx.save_for_backward(input, weight, mask, weight_g)
    ctx.has_bias = bias is not None


def _ml_bwd(ctx, grad_output):
    input, weight, mask, weight_g = ctx.saved_tensors
    gi, gw, gb, gg = torch.ops.tfep.masked_linear_backward(grad_output.contiguous(), input, weight, mask, weight_g)
    return gi, gw, (gb if ctx.has_bias else None), None, (gg if weight_g is not None else None)


masked_linear.register_autograd(_ml_bwd, setup_context=_ml_setup)


# ============================================================================= fused MADE output layer + transformer

def _fused_launch(h, h_inv_scale, w, w_inv_scale, bias, k_ranges, tile_order, kind, x, y, feat_index, feat_tr, n_slots,
                  n_rows, x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound, learn_upper_bound,
                  min_bin_size, min_slope):
    x, ldx = _lib.rows(x, 'x')
    B, D = x.shape
    if y.shape != x.shape or y.stride(-1) != 1 or (B > 1 and y.stride(0) != D):
        raise RuntimeError('fused_output_transformer: y must be a contiguous tensor of the shape of x')
    ldj = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = torch.empty(n_slots // 16, B, dtype=torch.float64, device=x.device)
    desc = None
    if kind == 1:
        desc = _spline_cfg(x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound,
                           learn_upper_bound, min_bin_size, min_slope).desc
    tail = (_lib.ptr(bias), _lib.ptr(k_ranges), _lib.ptr(tile_order), kind,
            ctypes.byref(desc) if desc is not None else None,
            _lib.ptr(x), ldx, _lib.ptr(y), D, _lib.ptr(feat_index), _lib.ptr(feat_tr),
            n_slots, _lib.ptr(ws), _lib.ptr(ldj), 0, B, n_rows, w.shape[1], _lib.stream_of(x))
    if h_inv_scale is not None:
        _lib.call('tfep_fused_output_transformer_forward_split', _lib.ptr(h), h.shape[1], _lib.ptr(h_inv_scale),
                  _lib.ptr(w), w.shape[1], _lib.ptr(w_inv_scale), *tail)
    else:
        _lib.call('tfep_fused_output_transformer_forward', _lib.ptr(h), h.shape[1], _lib.ptr(w), w.shape[1], *tail)
    return ldj


@custom_op('tfep::fused_output_transformer', mutates_args=(), device_types=_DEV)
def fused_output_transformer(h: Tensor, h_inv_scale: Optional[Tensor], w: Tensor, w_inv_scale: Optional[Tensor],
                             bias: Tensor, k_ranges: Tensor, tile_order: Tensor, kind: int, x: Tensor, y_init: Optional[Tensor],
                             feat_index: Tensor, feat_tr: Tensor, n_slots: int, n_rows: int,
                             x0: Optional[Tensor], xf: Optional[Tensor], y0: Optional[Tensor], yf: Optional[Tensor],
                             n_bins: int, circular: bool, identity_boundary_slopes: bool, learn_lower_bound: bool,
                             learn_upper_bound: bool, min_bin_size: float, min_slope: float) -> Tuple[Tensor, Tensor]:
    """``tfep_fused_output_transformer_forward[_split]``: the MADE output-layer GEMM with the affine (kind 0) or RQ-spline
    (kind 1) transformer and the log-det in its epilogue.  ``h`` / ``w``: last hidden activations and packed output
    weights -- split-f16 rows when ``h_inv_scale`` / ``w_inv_scale`` are given, fp32 otherwise.  ``y_init``: the input
    with its fixed features (copied through), or None when every feature is transformed."""
    y = y_init.clone() if y_init is not None else torch.empty(x.shape, dtype=x.dtype, device=x.device)
    ldj = _fused_launch(h, h_inv_scale, w, w_inv_scale, bias, k_ranges, tile_order, kind, x, y, feat_index, feat_tr, n_slots,
                        n_rows, x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound,
                        learn_upper_bound, min_bin_size, min_slope)
    return y, ldj


@fused_output_transformer.register_fake
def _(h, h_inv_scale, w, w_inv_scale, bias, k_ranges, tile_order, kind, x, y_init, *rest):
    return x.new_empty(x.shape), x.new_empty((x.shape[0],))


@custom_op('tfep::fused_output_transformer_', mutates_args=('y',), device_types=_DEV)
def fused_output_transformer_(h: Tensor, h_inv_scale: Optional[Tensor], w: Tensor, w_inv_scale: Optional[Tensor],
                              bias: Tensor, k_ranges: Tensor, tile_order: Tensor, kind: int, x: Tensor, y: Tensor,
                              feat_index: Tensor, feat_tr: Tensor, n_slots: int, n_rows: int,
                              x0: Optional[Tensor], xf: Optional[Tensor], y0: Optional[Tensor], yf: Optional[Tensor],
                              n_bins: int, circular: bool, identity_boundary_slopes: bool, learn_lower_bound: bool,
                              learn_upper_bound: bool, min_bin_size: float, min_slope: float) -> Tensor:
    """The same launch writing the columns ``feat_index`` of an existing ``y`` (in place) and returning the log-det of
    those features: the groups of a mixed transformer are one launch each on their rows of the packed weights."""
    return _fused_launch(h, h_inv_scale, w, w_inv_scale, bias, k_ranges, tile_order, kind, x, y, feat_index, feat_tr, n_slots,
                         n_rows, x0, xf, y0, yf, n_bins, circular, identity_boundary_slopes, learn_lower_bound,
                         learn_upper_bound, min_bin_size, min_slope)


@fused_output_transformer_.register_fake
def _(h, h_inv_scale, w, w_inv_scale, bias, k_ranges, tile_order, kind, x, y, *rest):
    return x.new_empty((x.shape[0],))


# ============================================================================= TFEP reductions

@custom_op('tfep::tfep_reduce', mutates_args=(), device_types=_DEV)
def tfep_reduce(target_potentials: Tensor, log_det_J: Optional[Tensor], ref_potentials: Optional[Tensor],
                log_weights: Optional[Tensor], bias: Optional[Tensor], kT: float, ignore_nan: bool) -> Tensor:
    return ops.tfep_reduce(target_potentials, log_det_J, ref_potentials, log_weights, bias, kT=kT, ignore_nan=ignore_nan)


@tfep_reduce.register_fake
def _(target_potentials, log_det_J, ref_potentials, log_weights, bias, kT, ignore_nan):
    return target_potentials.new_empty((9,), dtype=torch.float64)


OPS = ('affine_forward', 'affine_inverse', 'affine_backward', 'spline_forward', 'spline_inverse', 'spline_backward',
       'moebius_forward', 'moebius_inverse', 'moebius_backward', 'masked_linear', 'masked_linear_backward',
       'fused_output_transformer', 'fused_output_transformer_', 'tfep_reduce')
