"""Tensor-level wrappers of the C ABI (one function per entry point of include/tfep_hip.h).

Every function takes float32 HIP tensors, launches on the current HIP stream of the
tensor's device and returns fresh output tensors (inputs are never modified, like the
reference: flows/autoregressive.py:165-166).  No autograd, no CPU path.
"""
import ctypes
import os

import torch

from . import _lib
from ._lib import ParamLayout, SplineDesc, call, check_device_tensor, ptr, rows, stream_of


def _ldj_out(log_det_J, B, like):
    """Return (tensor, accumulate flag): accumulate into ``log_det_J`` if given."""
    if log_det_J is None:
        return torch.empty(B, dtype=torch.float32, device=like.device), 0
    check_device_tensor(log_det_J, 'log_det_J')
    if log_det_J.shape != (B,) or not log_det_J.is_contiguous():
        raise ValueError('log_det_J must be a contiguous (batch,) tensor')
    return log_det_J, 1


def _check_params(parameters, B, n, name='parameters'):
    parameters, ld = rows(parameters, name)
    if parameters.shape[0] != B or parameters.shape[1] != n:
        raise ValueError(f'{name} must have shape ({B}, {n}), got {tuple(parameters.shape)}')
    return parameters, ld


def _layout(ld, D, layout=None):
    if layout is not None:
        return ParamLayout(*layout)
    return ParamLayout(ld, D, 1)                          # reference layout: column p*D + f


# ----------------------------------------------------------------------------- affine

def affine(x, parameters, inverse=False, log_det_J=None):
    """AffineTransformer.forward / .inverse (reference affine.py:51-106)."""
    x, ldx = rows(x, 'x')
    B, D = x.shape
    parameters, ldp = _check_params(parameters, B, 2 * D)
    y = torch.empty(B, D, dtype=x.dtype, device=x.device)
    ldj, acc = _ldj_out(log_det_J, B, x)
    fn = 'tfep_affine_inverse' if inverse else 'tfep_affine_forward'
    call(fn, ptr(x), ldx, ptr(parameters), _layout(ldp, D), ptr(y), max(D, 1), ptr(ldj), acc, B, D, stream_of(x))
    return y, ldj


def volume_preserving_shift(x, shift, periodic_mask=None, limits=(0.0, 1.0), inverse=False):
    """VolumePreservingShiftTransformer (reference affine.py:366-456); log-det is zero."""
    x, ldx = rows(x, 'x')
    B, D = x.shape
    shift, lds = _check_params(shift, B, D, 'shift')
    y = torch.empty(B, D, dtype=x.dtype, device=x.device)
    if periodic_mask is not None:
        check_device_tensor(periodic_mask, 'periodic_mask', torch.int32)
    call('tfep_volume_preserving_shift', ptr(x), ldx, ptr(shift), lds,
         ptr(periodic_mask), float(limits[0]), float(limits[1]), -1 if inverse else 1,
         ptr(y), max(D, 1), B, D, stream_of(x))
    return y, zeros(B, dtype=x.dtype, device=x.device)


# ----------------------------------------------------------------------------- spline

class SplineConfig:
    """Host mirror of tfep_spline_desc; keeps the (D,) device arrays alive."""

    def __init__(self, x0, xf, y0, yf, n_bins, circular=False, identity_boundary_slopes=False,
                 learn_lower_bound=False, learn_upper_bound=False, min_bin_size=1e-4, min_slope=1e-4):
        self.x0, self.xf, self.y0, self.yf = (check_device_tensor(t.contiguous(), n)
                                              for t, n in ((x0, 'x0'), (xf, 'xf'), (y0, 'y0'), (yf, 'yf')))
        self.desc = SplineDesc(self.x0.data_ptr(), self.xf.data_ptr(), self.y0.data_ptr(), self.yf.data_ptr(),
                               int(n_bins), int(bool(circular)), int(bool(identity_boundary_slopes)),
                               int(bool(learn_lower_bound)), int(bool(learn_upper_bound)),
                               float(min_bin_size), float(min_slope))
        self.n_parameters_per_feature = _lib.load().tfep_spline_n_parameters_per_feature(ctypes.byref(self.desc))


def spline(x, parameters, cfg, inverse=False, log_det_J=None, layout=None):
    """NeuralSplineTransformer.forward / .inverse (reference spline.py:184-261)."""
    x, ldx = rows(x, 'x')
    B, D = x.shape
    P = cfg.n_parameters_per_feature
    if cfg.x0.numel() != D:
        raise ValueError(f'spline domain has {cfg.x0.numel()} features, input has {D}')
    parameters, ldp = _check_params(parameters, B, P * D)
    y = torch.empty(B, D, dtype=x.dtype, device=x.device)
    ldj, acc = _ldj_out(log_det_J, B, x)
    fn = 'tfep_spline_inverse' if inverse else 'tfep_spline_forward'
    call(fn, ptr(x), ldx, ptr(parameters), _layout(ldp, D, layout), ctypes.byref(cfg.desc),
         ptr(y), max(D, 1), ptr(ldj), acc, B, D, stream_of(x))
    return y, ldj


# ----------------------------------------------------------------------------- moebius

def moebius(x, parameters, dimension, max_radius=0.99, unit_sphere=False, inverse=False, log_det_J=None):
    """MoebiusTransformer.forward / .inverse (reference moebius.py:104-147, :374-478)."""
    x, ldx = rows(x, 'x')
    B, D = x.shape
    parameters, ldp = _check_params(parameters, B, D)
    y = torch.empty(B, D, dtype=x.dtype, device=x.device)
    ldj, acc = _ldj_out(log_det_J, B, x)
    call('tfep_moebius_forward', ptr(x), ldx, ptr(parameters), ldp,
         int(dimension), float(max_radius), int(bool(unit_sphere)), -1 if inverse else 1,
         ptr(y), max(D, 1), ptr(ldj), acc, B, D, stream_of(x))
    return y, ldj


def moebius_split_out(x, parameters, max_radius, cols_padded):
    """Forward map of unit-sphere 2-vectors that also returns y as split-f16 rows ``(y, log_det_J, y_split, y_inv_scale)`` for
    the next masked linear (``tfep_moebius_forward_split_out``)."""
    x, ldx = rows(x, 'x')
    B, D = x.shape
    parameters, ldp = _check_params(parameters, B, D)
    if ldx % 2 or ldp % 2 or x.data_ptr() % 8 or parameters.data_ptr() % 8:
        x, ldx, parameters, ldp = x.contiguous(), D, parameters.contiguous(), D
    y = torch.empty(B, D, dtype=x.dtype, device=x.device)
    ldj = torch.empty(B, dtype=x.dtype, device=x.device)
    ys = zeros(B, cols_padded, dtype=torch.float32, device=x.device) if cols_padded > D else \
        torch.empty(B, cols_padded, dtype=torch.float32, device=x.device)
    ys_inv = torch.empty(max(B, 1), dtype=torch.float32, device=x.device)
    call('tfep_moebius_forward_split_out', ptr(x), ldx, ptr(parameters), ldp, float(max_radius), ptr(y), D, ptr(ldj), 0, ptr(ys),
         cols_padded, ptr(ys_inv), B, D, stream_of(x))
    return y, ldj, ys, ys_inv


# ----------------------------------------------------------------------------- embedding / index helpers

def periodic_embedding(x, periodic_indices, nonperiodic_indices, lower, upper):
    """PeriodicEmbedding.forward (reference mafembed.py:112-145)."""
    x, ldx = rows(x, 'x')
    B = x.shape[0]
    n_per, n_non = periodic_indices.numel(), nonperiodic_indices.numel()
    out = torch.empty(B, n_non + 2 * n_per, dtype=x.dtype, device=x.device)
    call('tfep_periodic_embedding', ptr(x), ldx, ptr(periodic_indices), n_per, ptr(nonperiodic_indices), n_non,
         float(lower), float(upper), ptr(out), n_non + 2 * n_per, B, stream_of(x))
    return out


def gather_columns(src, idx):
    src, lds = rows(src, 'src')
    B, n = src.shape[0], idx.numel()
    dst = torch.empty(B, n, dtype=src.dtype, device=src.device)
    call('tfep_gather_columns', ptr(src), lds, ptr(idx), n, ptr(dst), n, B, stream_of(src))
    return dst


def scatter_columns(src, idx, dst):
    """dst[:, idx[j]] = src[:, j] (in place on ``dst``, which the caller owns)."""
    src, lds = rows(src, 'src')
    B, n = src.shape[0], idx.numel()
    if not dst.is_contiguous():
        raise ValueError('dst must be contiguous')
    call('tfep_scatter_columns', ptr(src), lds, ptr(idx), n, ptr(dst), dst.shape[1], B, stream_of(src))
    return dst


# ----------------------------------------------------------------------------- masked linear

_TILES = {}


def tile_sizes():
    if 'wide' not in _TILES:                    # (constants of the library: asked once, not on every launch)
        lib = _lib.load()
        _TILES['wide'] = (lib.tfep_masked_linear_tile_m(), lib.tfep_masked_linear_tile_n(), lib.tfep_masked_linear_tile_k())
    return _TILES['wide']


def round_up(n, m):
    return (n + m - 1) // m * m


def pad_columns(x, k_padded):
    """Return ``x`` as a (B, k_padded) buffer with zero padding (GEMM operand contract)."""
    x, _ = rows(x, 'x')
    B, K = x.shape
    if K == k_padded and x.stride(0) == K and x.data_ptr() % 16 == 0:
        return x
    out = zeros(B, k_padded, dtype=x.dtype, device=x.device)
    out[:, :K] = x
    return out


def masked_weight_prepare(weight_v, weight_g=None, mask=None, row_of_out=None, col_of_in=None,
                          n_rows_padded=None, k_padded=None, out=None, col_cut=None, clear=True, in_of_col=None):
    """Effective masked weight, permuted + zero padded (reference masked.py:369-371, :433-439, :270).  ``col_cut``:
    prefix mask rows (see ``masked_weight_prepare_split``); ``clear=False``: ``out`` was zeroed once and always holds the
    same layer, so its padding needs no clearing.  With ``col_cut``, ``clear=False`` and ``in_of_col`` (the inverse of
    ``col_of_in``) long rows take the LDS-staged prefix kernel, which writes only the live prefix of each packed row."""
    check_device_tensor(weight_v, 'weight')
    N, K = weight_v.shape
    tk = tile_sizes()[2]
    n_rows_padded = N if n_rows_padded is None else n_rows_padded
    k_padded = round_up(K, tk) if k_padded is None else k_padded
    if out is None:
        out = torch.empty(n_rows_padded, k_padded, dtype=torch.float32, device=weight_v.device)
    # (contiguous copies live in locals until the launch is queued; a temporary freed inside the argument list could be
    # overwritten by the next one)
    v_c = weight_v.contiguous()
    g_c = None if weight_g is None else weight_g.contiguous()
    m_c = None if mask is None else mask.contiguous()
    if col_cut is not None and not clear and (in_of_col is not None or col_of_in is None) and 8192 <= K <= 16384 and \
            out.shape[1] % 4 == 0 and out.shape[1] >= round_up(K, 8) and os.environ.get('TFEP_PACK_LDS', '1') != '0':
        call('tfep_masked_weight_prepare_prefix', ptr(v_c), ptr(g_c), N, K, ptr(row_of_out), ptr(in_of_col), ptr(col_cut),
             ptr(out), n_rows_padded, out.shape[1], stream_of(weight_v))
        return out
    call('tfep_masked_weight_prepare', ptr(v_c), ptr(g_c), ptr(m_c), N, K, ptr(row_of_out), ptr(col_of_in), ptr(col_cut),
         int(bool(clear)), ptr(out), n_rows_padded, k_padded, stream_of(weight_v))
    return out


def mask_k_ranges(mask, tile_n, n_tiles, k_padded, row_of_out=None, col_of_in=None):
    check_device_tensor(mask, 'mask')
    N, K = mask.shape
    out = torch.empty(n_tiles, 2, dtype=torch.int32, device=mask.device)
    m_c = mask.contiguous()
    call('tfep_mask_k_ranges', ptr(m_c), N, K, ptr(row_of_out), ptr(col_of_in), tile_n,
         tile_sizes()[2], n_tiles, k_padded, ptr(out), stream_of(mask))
    return out


def heavy_first_order(k_ranges):
    """Column tiles sorted by descending k-range length (host-side, once per mask)."""
    kr = k_ranges.cpu().long()
    order = torch.argsort(kr[:, 1] - kr[:, 0], descending=True, stable=True)
    return order.to(device=k_ranges.device, dtype=torch.int32)


def xcd_balanced_tile_list(live, n_xcd=8, group=32):
    """Launch order of the live output tiles of a block-sparse product (``tile_list`` of ``tfep_gemm_desc``).

    ``live``: (row tiles, column tiles) bool / uint8 CPU tensor.  Workgroup ids go round-robin to the 8 XCDs and each XCD
    works through its own share, so the live tiles are listed in the order of the 8 x 4 super-tile walk (neighbours share
    operand panels in the XCD's L2), cut into groups of 32, and group k is placed at the positions of XCD k % 8: every XCD
    gets the same number of tiles whatever the shape of the live region.  Returns an int32 (n_positions, 2) CPU tensor,
    -1 = no tile."""
    live = torch.as_tensor(live).cpu().bool()
    mt, nt = torch.nonzero(live, as_tuple=True)
    # order of the 8 x 4 super-tile walk: column super-tile, row super-tile, then column-major inside the super-tile
    key = (((nt // 4) * ((live.shape[0] + 7) // 8) + mt // 8) * 4 + nt % 4) * 8 + mt % 8
    order = torch.argsort(key)
    mt, nt = mt[order], nt[order]
    n = mt.numel()
    n_groups = (n + group - 1) // group
    n_rounds = max(1, (n_groups + n_xcd - 1) // n_xcd)
    out = torch.full((n_rounds * group * n_xcd, 2), -1, dtype=torch.int32)
    i = torch.arange(n)
    k, w = i // group, i % group
    pos = ((k // n_xcd) * group + w) * n_xcd + k % n_xcd
    out[pos, 0], out[pos, 1] = mt.to(torch.int32), nt.to(torch.int32)
    return out


def split_wide_tile_n():
    """Widest column tile of the split-f16 kernel's plain linear product (400)."""
    if 'xwide' not in _TILES:
        _TILES['xwide'] = _lib.load().tfep_split_wide_tile_n()
    return _TILES['xwide']


def narrow_tile_n():
    if 'narrow' not in _TILES:
        _TILES['narrow'] = _lib.load().tfep_masked_linear_narrow_tile_n()
    return _TILES['narrow']


def few_wide_tiles(B, N):
    """True when a (B x N) product is at most 128 of the 256 x 256 tiles, half the CUs (and more than one 32-column tile wide):
    one workgroup per wide tile then leaves CUs idle and the run time is one workgroup's walk over k, see
    ``masked_linear_packed``.  Measured (tools/probe/tile_choice.py, exact-fp32 kernels, dense): B 1024 x K 800 x N 800
    (16 wide tiles) 214 us wide / 42 us narrow; 4096 x 800 x 800 (64) 221 / 73; 8192 x 800 x 800 (128) 227 / 119;
    4096 x 800 x 3200 (208) 236 / 207."""
    tm, tn, _ = tile_sizes()
    # (the narrow tile runs dense -- the mask tables are per wide tile -- so at ~200 tiles, where the two are level on a
    # dense product, the wide tile with its k-ranges wins: the line is drawn at half the CUs)
    return ((B + tm - 1) // tm) * ((N + tn - 1) // tn) <= 128 and N > narrow_tile_n()


def masked_linear_packed(x_padded, w_packed, bias, n_out, k_ranges=None, col_map=None, act=0, out=None,
                         out_cols=None, tile_order=None):
    """y = act(x W^T + b) on packed operands (reference masked.py:265-277 + made.py:320).  Small products (cfg1-sized
    layers: a handful of 256 x 256 tiles) run on the 32-column tile instead, dense: more workgroups, shorter chains."""
    B = x_padded.shape[0]
    n_rows_w, k_padded = w_packed.shape
    if out is None:
        out = torch.empty(B, n_out if out_cols is None else out_cols, dtype=torch.float32, device=x_padded.device)
    tile_n = 0
    if col_map is None and few_wide_tiles(B, n_out):
        tile_n, k_ranges, tile_order = narrow_tile_n(), None, None      # (the mask tables are per 256-column tile)
    call('tfep_masked_linear_forward', ptr(x_padded), x_padded.shape[1], ptr(w_packed), k_padded,
         ptr(bias), ptr(k_ranges), ptr(tile_order), ptr(col_map), ptr(out), out.shape[1], B, n_out, n_rows_w,
         k_padded, int(act), tile_n,
         stream_of(x_padded))
    return out


def gemm_slice(x_padded, w_packed, row0, n_rows, bias, k_ranges, kr_offset, out, col0, act):
    """Row slice of a packed masked linear layer with the narrow column tile:
    ``out[:, col0:col0+n_rows] = act(x W[row0:row0+n_rows]^T + bias[row0:row0+n_rows])``.
    ``k_ranges[kr_offset:]`` holds the [k_begin, k_end) of the slice's 32-row tiles."""
    B = x_padded.shape[0]
    k_padded = w_packed.shape[1]
    esz = 4
    call('tfep_masked_linear_forward', ptr(x_padded), x_padded.shape[1],
         ctypes.c_void_p(w_packed.data_ptr() + row0 * k_padded * esz), k_padded,
         ctypes.c_void_p(bias.data_ptr() + row0 * esz),
         ctypes.c_void_p(k_ranges.data_ptr() + kr_offset * 2 * 4), None, None,
         ctypes.c_void_p(out.data_ptr() + col0 * esz), out.shape[1], B, n_rows, n_rows, k_padded, int(act),
         _lib.load().tfep_masked_linear_narrow_tile_n(), stream_of(x_padded))


# ----------------------------------------------------------------------------- split-precision operands

def split_gemm_enabled():
    """The MADE GEMMs of the forward pass run on split-f16 operands (3 fp16 MFMAs per fp32 product, fp32-equivalent
    results) unless ``TFEP_SPLIT_GEMM=0`` selects the exact-fp32 MFMA kernel."""
    return os.environ.get('TFEP_SPLIT_GEMM', '1') != '0'


def split_rows(x, cols_padded, per_tensor=False, out=None, inv_scale=None):
    """fp32 rows -> split-f16 rows (``tfep_split_rows``).  Returns ``(split, inv_scale)``: ``split`` is a
    (rows, cols_padded) float32-TYPED container of the bit pattern, ``inv_scale`` has one entry per row, or
    [1/scale, scratch] with ``per_tensor``."""
    x, ldx = rows(x, 'x')
    R, C = x.shape
    if out is None:
        out = torch.empty(R, cols_padded, dtype=torch.float32, device=x.device)
    if inv_scale is None:
        inv_scale = torch.empty(2 if per_tensor else max(R, 1), dtype=torch.float32, device=x.device)
    call('tfep_split_rows', ptr(x), ldx, R, C, ptr(out), out.shape[1], cols_padded, ptr(inv_scale), int(per_tensor),
         stream_of(x))
    return out, inv_scale


def zeros(*shape, dtype=torch.float32, device=None):
    """``torch.zeros`` as a fill KERNEL.  ``torch.zeros`` clears with ``hipMemsetAsync``; captured in a HIP graph that becomes
    a memset node, and memset nodes were seen to leave garbage behind on replay (ROCm 7.0 / MI355X: the padding of a packed
    weight buffer cleared by ``hipMemsetAsync`` read back as ~1e36 by a later kernel of the same graph while eager runs
    were clean -- the replayed blocked inverse differed from the eager one from the second block on; earlier, a 4-byte
    memset was corrupted by a second capture).  Everything on a capturable path clears its buffers with kernels."""
    return torch.full(tuple(shape), 0.0, dtype=dtype, device=device)


def split_columns_scaled(x, col0, cols, out, inv_scale):
    """Columns ``[col0, col0 + cols)`` (whole groups of 8; ``col0 % 8 == 0``) of the fp32 rows ``x`` into the same columns
    of the split rows ``out``, with the caller's per-row ``inv_scale`` (powers of two): for operands filled incrementally
    (``tfep_split_columns_scaled``)."""
    x, ldx = rows(x, 'x')
    call('tfep_split_columns_scaled', ptr(x), ldx, x.shape[0], int(col0), int(cols), ptr(out), out.shape[1], ptr(inv_scale),
         stream_of(x))
    return out


def column_absmax(x):
    """``max_r |x[r, c]|`` per column of a 2-D tensor as a (1, cols) float32 tensor (``tfep_column_absmax``; NaN survives)."""
    x, ldx = rows(x, 'x')
    out = torch.empty(1, x.shape[1], dtype=torch.float32, device=x.device)
    call('tfep_column_absmax', ptr(x), ldx, x.shape[0], x.shape[1], ptr(out), stream_of(x))
    return out


def range_flag(tensors, bits=19):
    """Number of rows, over the 2-D fp32 ``tensors``, whose own dynamic range exceeds what the split-f16 operand format
    carries at fp32 accuracy (a non-zero element below ``2^-bits`` of the row maximum, or a non-finite element):
    ``tfep_range_flag`` into one device counter, read back once (ONE host synchronisation for the whole list)."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not tensors:
        return 0
    count = torch.zeros(1, dtype=torch.int32, device=tensors[0].device)
    for t in tensors:
        check_device_tensor(t, 'range_flag input')
        if t.dim() != 2 or t.stride(1) != 1:
            t = t.reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.reshape(1, -1).contiguous()
        call('tfep_range_flag', ptr(t), t.stride(0) if t.shape[0] > 1 else t.shape[1], t.shape[0], t.shape[1], int(bits),
             ptr(count), stream_of(t))
    return int(count.item())


def range_flag_device(tensors, bits=19, count=None):
    """``range_flag`` without the read-back: the device counter (int32, 1 element) itself, a new one or ``count`` added to --
    capturable in a HIP graph (a new counter is cleared by a fill kernel, see ``zeros``).  ``None`` when there is nothing to check."""
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not tensors:
        return count
    if count is None:
        count = zeros(1, dtype=torch.int32, device=tensors[0].device)
    for t in tensors:
        check_device_tensor(t, 'range_flag input')
        if t.dim() != 2 or t.stride(1) != 1:
            t = t.reshape(t.shape[0], -1).contiguous() if t.dim() > 1 else t.reshape(1, -1).contiguous()
        call('tfep_range_flag', ptr(t), t.stride(0) if t.shape[0] > 1 else t.shape[1], t.shape[0], t.shape[1], int(bits),
             ptr(count), stream_of(t))
    return count


def abs_reduce(x, what):
    """``what='row_max'``: max_k |x[row, k]| per row;  ``'max_row_sum'``: max_row sum_k |x[row, k]| as a 1-element tensor
    (``tfep_abs_reduce``: plain kernels, capturable in a HIP graph, unlike torch's multi-block reductions whose semaphores are
    cleared with a small memset)."""
    x, ldx = rows(x, 'x')
    mode = {'row_max': 0, 'max_row_sum': 1}[what]
    out = torch.empty(x.shape[0] if mode == 0 else 1, dtype=torch.float32, device=x.device)
    call('tfep_abs_reduce', ptr(x), ldx, x.shape[0], x.shape[1], mode, ptr(out), stream_of(x))
    return out


def pow2_inv_scale(bound):
    """1 / s for the power of two s that puts ``bound`` (> 0, per row) into [2^14, 2^15): the inverse scale of a split row
    whose entries are known to stay below ``bound`` (the split format keeps an absolute error of 2^-40 of the scaled
    maximum, so a loose bound costs nothing)."""
    e = torch.floor(torch.log2(bound.clamp_min(1e-30).double()))
    return torch.exp2(e - 14.0).float()


def masked_weight_prepare_split(weight_v, weight_g, mask, row_of_out, in_of_col, out, inv_scale, col_cut=None):
    """Effective masked weight written directly as split-f16 rows into ``out`` (n_rows_padded, k_padded), whose
    padding rows must already be zero (``tfep_masked_weight_prepare_split``).  ``inv_scale``: 4 floats --
    [1/scale, scratch, max_j sum_k |w_jk|, unused].  ``col_cut`` (int32 per output row): the mask rows are prefixes of the
    packed columns, ``mask[o, in_of_col[c]] == (c < col_cut[o])`` -- the mask is then not read."""
    check_device_tensor(weight_v, 'weight')
    if inv_scale.numel() < 4:
        raise ValueError('inv_scale must have 4 entries')
    N, K = weight_v.shape
    v_c = weight_v.contiguous()
    g_c = None if weight_g is None else weight_g.contiguous()
    m_c = None if mask is None else mask.contiguous()
    call('tfep_masked_weight_prepare_split', ptr(v_c), ptr(g_c), ptr(m_c), N, K, ptr(row_of_out), ptr(in_of_col), ptr(col_cut),
         ptr(out), out.shape[1], out.shape[1], ptr(inv_scale), stream_of(weight_v))
    return out, inv_scale


def masked_linear_split(x_split, x_inv_scale, w_split, w_inv_scale, bias, n_out, k_ranges=None, act=0, out=None,
                        tile_order=None, split_out=False, bias_absmax=None, k_split=1):
    """``masked_linear_packed`` on split-f16 operands; the output is ordinary fp32, or -- ``split_out`` with
    ``act=1`` -- the ELU activations as split rows for the next layer: returns ``(rows, inv_scale)`` then.
    ``w_inv_scale`` is the 4-float buffer of ``masked_weight_prepare_split`` (its entry 2 bounds the outputs).
    ``k_split`` > 1 (plain linear product only): the k range is cut into that many slices whose partial sums go to the
    slabs ``out[s]`` of an ``(k_split, B, n_out)`` output (bias in slab 0); the caller adds them."""
    B = x_split.shape[0]
    n_rows_w, k_padded = w_split.shape
    if out is None:
        out = torch.empty((B, n_out) if k_split <= 1 else (k_split, B, n_out), dtype=torch.float32, device=x_split.device)
    d = _lib.GemmDesc()
    if split_out:
        if bias_absmax is None:
            bias_absmax = abs_reduce(bias.reshape(1, -1), 'row_max') if bias is not None else zeros(1, device=out.device)
        y_inv = torch.empty(max(B, 1), dtype=torch.float32, device=out.device)
        d.split_out, d.y_inv_scale = 1, y_inv.data_ptr()
        d.w_l1max, d.bias_absmax = w_inv_scale.data_ptr() + 8, bias_absmax.data_ptr()
    d.x, d.ldx = x_split.data_ptr(), x_split.shape[1]
    d.w, d.ldw = w_split.data_ptr(), k_padded
    d.bias = bias.data_ptr() if bias is not None else None
    d.k_ranges = k_ranges.data_ptr() if k_ranges is not None else None
    d.tile_order = tile_order.data_ptr() if tile_order is not None else None
    d.col_map = None
    d.y, d.ldy = out.data_ptr(), out.shape[-1]
    d.B, d.N, d.n_rows_w, d.k_padded, d.act, d.accumulate = B, n_out, n_rows_w, k_padded, int(act), 0
    d.split, d.x_inv_scale, d.w_inv_scale = 1, x_inv_scale.data_ptr(), w_inv_scale.data_ptr()
    if k_split > 1:
        d.k_split, d.slab_stride = int(k_split), out.shape[-2] * out.shape[-1]
    call('tfep_masked_linear_gemm', ctypes.byref(d), stream_of(x_split))
    return (out, y_inv) if split_out else out


# ----------------------------------------------------------------------------- reductions

def tfep_reduce(target_potentials, log_det_J=None, ref_potentials=None, log_weights=None, bias=None,
                kT=1.0, ignore_nan=False):
    """The 9 float64 sufficient statistics of the TFEP loss / estimator (see tfep_hip.h)."""
    t = check_device_tensor(target_potentials.contiguous(), 'target_potentials')
    N = t.numel()
    opt = []
    for v, n in ((log_det_J, 'log_det_J'), (ref_potentials, 'ref_potentials'), (log_weights, 'log_weights'),
                 (bias, 'bias')):
        if v is not None:
            v = check_device_tensor(v.contiguous(), n)
            if v.numel() != N:
                raise ValueError(f'{n} must have {N} elements')
        opt.append(v)
    nws = _lib.load().tfep_tfep_reduce_workspace_doubles(N)
    ws = torch.empty(nws, dtype=torch.float64, device=t.device)
    out = torch.empty(9, dtype=torch.float64, device=t.device)
    call('tfep_tfep_reduce', ptr(t), ptr(opt[0]), ptr(opt[1]), ptr(opt[2]), ptr(opt[3]), float(kT),
         int(bool(ignore_nan)), N, ptr(ws), ptr(out), stream_of(t))
    return out
