"""ctypes binding of libtfep_hip.so (the C ABI declared in include/tfep_hip.h).

There is NO CPU fallback: if the library is missing or a tensor is not a float32
HIP tensor, calls raise.  ``import torch`` happens first so that the library binds
to the HIP runtime PyTorch-ROCm already loaded (same ``libamdhip64.so.7`` soname),
which is what makes ``tensor.data_ptr()`` and the current stream valid on our side.
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('TFEP_HIP_LIB') or os.path.join(_HERE, 'lib', 'libtfep_hip.so')
ABI_VERSION = 8

_lib = None


class ParamLayout(Structure):
    _fields_ = [('ld', c_int64), ('stride_p', c_int64), ('stride_f', c_int64)]


class GemmDesc(Structure):
    _fields_ = [('x', c_void_p), ('ldx', c_int64), ('w', c_void_p), ('ldw', c_int64), ('bias', c_void_p),
                ('k_ranges', c_void_p), ('tile_order', c_void_p), ('col_map', c_void_p),
                ('y', c_void_p), ('ldy', c_int64),
                ('B', c_int32), ('N', c_int32), ('n_rows_w', c_int32), ('k_padded', c_int32),
                ('act', c_int32), ('accumulate', c_int32),
                ('elu_grad_of', c_void_p), ('ld_elu_grad_of', c_int64), ('tile_live', c_void_p),
                ('pre_add', c_void_p), ('ld_pre_add', c_int64), ('tile_n', c_int32),
                ('split', c_int32), ('x_inv_scale', c_void_p), ('w_inv_scale', c_void_p),
                ('split_out', c_int32), ('y_inv_scale', c_void_p), ('w_l1max', c_void_p), ('bias_absmax', c_void_p),
                ('k_split', c_int32), ('slab_stride', c_int64),
                ('tile_list', c_void_p), ('n_tile_list', c_int32)]


class InverseBlockDesc(Structure):
    _fields_ = [('B', c_int32), ('n_layers', c_int32), ('n_steps', c_int32), ('kind', c_int32),
                ('x', c_void_p), ('ldx', c_int64), ('xpad', c_void_p), ('ldxpad', c_int64),
                ('y', c_void_p), ('ldy', c_int64),
                ('h', c_void_p * 4), ('ldh', c_int64 * 4), ('z', c_void_p * 4), ('ldz', c_int64 * 4),
                ('zout', c_void_p), ('ldzout', c_int64),
                ('z_slabs', c_int32 * 4), ('z_slab_stride', c_int64 * 4),
                ('zout_slabs', c_int32), ('zout_slab_stride', c_int64), ('log_det_J', c_void_p),
                ('w', c_void_p * 4), ('ldw', c_int64 * 4), ('wout', c_void_p), ('ldwout', c_int64),
                ('steps', c_void_p), ('feat_cols', c_void_p), ('feat_sel', c_void_p),
                ('feat_in', c_void_p), ('feat_periodic', c_void_p), ('in_cols', c_void_p),
                ('emb_lower', c_float), ('emb_upper', c_float),
                ('cache_col0', c_int32 * 4), ('cache_n_old', c_int32 * 4),
                ('cache_len', c_int32), ('max_feats', c_int32), ('spline', c_void_p),
                ('moebius_dim', c_int32), ('moebius_unit_sphere', c_int32), ('moebius_max_radius', c_float),
                ('rows_per_wave', c_int32), ('n_spline_groups', c_int32), ('waves_per_workgroup', c_int32), ('paired', c_int32),
                ('n_blocks', c_int32), ('blocks', c_void_p), ('z_extra', c_void_p * 4), ('zout_extra', c_void_p),
                ('ws', c_void_p * 5), ('ldws', c_int64 * 5), ('ws_inv_scale', c_void_p * 5), ('h_inv_scale', c_void_p * 5)]


class MafLayerDesc(Structure):
    _fields_ = [('B', c_int32), ('n_linears', c_int32), ('a0', c_void_p), ('lda0', c_int64), ('a0_inv_scale', c_void_p),
                ('w', c_void_p * 4), ('ldw', c_int64 * 4), ('n_rows_w', c_int32 * 4), ('n_out', c_int32 * 4),
                ('w_scales', c_void_p * 4), ('bias', c_void_p * 4), ('bias_absmax', c_void_p * 4), ('k_ranges', c_void_p * 4),
                ('scratch', c_void_p * 2), ('ld_scratch', c_int64), ('kind', c_int32),
                ('x', c_void_p), ('ldx', c_int64), ('y', c_void_p), ('ldy', c_int64), ('log_det_J', c_void_p),
                ('n_features', c_int32), ('moebius_dim', c_int32), ('moebius_unit_sphere', c_int32),
                ('moebius_max_radius', c_float)]


class SplineDesc(Structure):
    _fields_ = [('x0', c_void_p), ('xf', c_void_p), ('y0', c_void_p), ('yf', c_void_p),
                ('n_bins', c_int32), ('circular', c_int32), ('identity_boundary_slopes', c_int32),
                ('learn_lower_bound', c_int32), ('learn_upper_bound', c_int32),
                ('min_bin_size', c_float), ('min_slope', c_float)]


class EgnnLayerParams(Structure):
    _fields_ = [('F', c_int32), ('G', c_int32)] + [(k, c_void_p) for k in (
        'dist_means', 'dist_log_gammas', 'msg0_w', 'msg0_b', 'msg2_w', 'msg2_b', 'att_w', 'att_b', 'ux0_w', 'ux0_b',
        'ux2_w', 'uh0_w', 'uh0_b', 'uh2_w', 'uh2_b')]


class EgnnEdgeArgs(Structure):
    _fields_ = [('B', c_int32), ('n_nodes', c_int32), ('nt', c_int32), ('split', c_int32), ('r_cutoff', c_float),
                ('speed_factor', c_float),
                ('packed', c_void_p), ('pos', c_void_p), ('dpos', c_void_p), ('P', c_void_p), ('Q', c_void_p),
                ('pq_bstride', c_int64), ('dP', c_void_p), ('dQ', c_void_p), ('pos_out', c_void_p), ('dpos_out', c_void_p),
                ('nm', c_void_p), ('dnm', c_void_p)]


class EgnnNodeArgs(Structure):
    _fields_ = [('B', c_int32), ('n_nodes', c_int32), ('nt', c_int32), ('packed', c_void_p), ('packed_next', c_void_p),
                ('h', c_void_p), ('h_bstride', c_int64), ('dh', c_void_p), ('nm', c_void_p), ('dnm', c_void_p),
                ('h_out', c_void_p), ('dh_out', c_void_p), ('P_out', c_void_p), ('Q_out', c_void_p),
                ('dP_out', c_void_p), ('dQ_out', c_void_p)]


class EgnnEdgeBwdArgs(Structure):
    _fields_ = [('B', c_int32), ('n_nodes', c_int32), ('nt', c_int32), ('split', c_int32), ('src_owned', c_int32),
                ('r_cutoff', c_float), ('speed_factor', c_float), ('packed', c_void_p), ('pos', c_void_p),
                ('P', c_void_p), ('Q', c_void_p), ('pq_bstride', c_int64), ('g_pos_out', c_void_p), ('g_nm', c_void_p),
                ('g_lane', c_void_p), ('g_pos', c_void_p)]


class EgnnNodeBwdArgs(Structure):
    _fields_ = [('B', c_int32), ('n_nodes', c_int32), ('nt', c_int32), ('packed', c_void_p), ('packed_next', c_void_p),
                ('h', c_void_p), ('h_bstride', c_int64), ('nm', c_void_p), ('g_h_next', c_void_p), ('g_P', c_void_p),
                ('g_Q', c_void_p), ('g_h', c_void_p), ('g_nm', c_void_p)]


_P = c_void_p
_SIGNATURES = {
    'tfep_hip_abi_version': (c_int, []),
    'tfep_last_error': (c_char_p, []),
    'tfep_masked_weight_prepare': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, c_int, _P, c_int, c_int64, _P]),
    'tfep_masked_weight_prepare_prefix': (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_int, c_int64, _P]),
    'tfep_mask_k_ranges': (c_int, [_P, c_int, c_int, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    'tfep_masked_linear_forward': (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, _P, _P, c_int64,
                                           c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    'tfep_masked_linear_narrow_tile_n': (c_int, []),
    'tfep_masked_linear_tile_n': (c_int, []),
    'tfep_split_wide_tile_n': (c_int, []),
    'tfep_split_half_wide_tile_n': (c_int, []),
    'tfep_masked_linear_tile_k': (c_int, []),
    'tfep_masked_linear_tile_m': (c_int, []),
    'tfep_affine_forward': (c_int, [_P, c_int64, _P, ParamLayout, _P, c_int64, _P, c_int, c_int, c_int, _P]),
    'tfep_affine_inverse': (c_int, [_P, c_int64, _P, ParamLayout, _P, c_int64, _P, c_int, c_int, c_int, _P]),
    'tfep_volume_preserving_shift': (c_int, [_P, c_int64, _P, c_int64, _P, c_float, c_float, c_int,
                                             _P, c_int64, c_int, c_int, _P]),
    'tfep_spline_n_parameters_per_feature': (c_int, [POINTER(SplineDesc)]),
    'tfep_spline_forward': (c_int, [_P, c_int64, _P, ParamLayout, POINTER(SplineDesc), _P, c_int64,
                                    _P, c_int, c_int, c_int, _P]),
    'tfep_spline_inverse': (c_int, [_P, c_int64, _P, ParamLayout, POINTER(SplineDesc), _P, c_int64,
                                    _P, c_int, c_int, c_int, _P]),
    'tfep_moebius_forward': (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, c_int, c_int,
                                     _P, c_int64, _P, c_int, c_int, c_int, _P]),
    'tfep_moebius_forward_split_out': (c_int, [_P, c_int64, _P, c_int64, c_float, _P, c_int64, _P, c_int, _P, c_int64, _P,
                                               c_int, c_int, _P]),
    'tfep_periodic_embedding': (c_int, [_P, c_int64, _P, c_int, _P, c_int, c_float, c_float,
                                        _P, c_int64, c_int, _P]),
    'tfep_gather_columns': (c_int, [_P, c_int64, _P, c_int, _P, c_int64, c_int, _P]),
    'tfep_scatter_columns': (c_int, [_P, c_int64, _P, c_int, _P, c_int64, c_int, _P]),
    'tfep_fused_tile_features': (c_int, []),
    'tfep_fused_supported': (c_int, [c_int, POINTER(SplineDesc)]),
    'tfep_fused_tile_columns': (c_int, [c_int, POINTER(SplineDesc)]),
    'tfep_fused_output_transformer_forward': (c_int, [_P, c_int64, _P, c_int64, _P, _P, _P, c_int,
                                                      POINTER(SplineDesc), _P, c_int64, _P, c_int64,
                                                      _P, _P, c_int, _P, _P, c_int,
                                                      c_int, c_int, c_int, _P]),
    'tfep_split_tile_k': (c_int, []),
    'tfep_masked_weight_prepare_split': (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P, c_int64, c_int, _P, _P]),
    'tfep_masked_weight_prepare_split_both': (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, c_int64, _P, c_int64, c_int, _P, _P]),
    'tfep_split_rows': (c_int, [_P, c_int64, c_int64, c_int64, _P, c_int64, c_int64, _P, c_int, _P]),
    'tfep_abs_reduce': (c_int, [_P, c_int64, c_int64, c_int64, c_int, _P, _P]),
    'tfep_range_flag': (c_int, [_P, c_int64, c_int64, c_int64, c_int, _P, _P]),
    'tfep_column_absmax': (c_int, [_P, c_int64, c_int64, c_int64, _P, _P]),
    'tfep_split_columns_scaled': (c_int, [_P, c_int64, c_int64, c_int64, c_int64, _P, c_int64, _P, _P]),
    'tfep_fused_output_transformer_forward_split': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P, c_int,
                                                            POINTER(SplineDesc), _P, c_int64, _P, c_int64,
                                                            _P, _P, c_int, _P, _P, c_int,
                                                            c_int, c_int, c_int, _P]),
    'tfep_fused_saving_supported': (c_int, [POINTER(SplineDesc)]),
    'tfep_fused_output_transformer_forward_split_saving': (c_int, [_P, c_int64, _P, _P, c_int64, _P, _P, _P, _P,
                                                                   POINTER(SplineDesc), _P, c_int64, _P, c_int64,
                                                                   _P, _P, c_int, _P, _P, c_int,
                                                                   c_int, c_int, c_int, c_int, _P, c_int64, _P]),
    'tfep_diag_split_mfma_peak': (c_int, [_P, c_int, c_int, _P]),
    'tfep_diag_split_cycles': (c_int, [_P]),
    'tfep_inverse_block_step_ints': (c_int, []),
    'tfep_inverse_block_record_ints': (c_int, []),
    'tfep_inverse_block_lds_bytes': (c_int64, [c_int, c_int, c_int]),
    'tfep_inverse_block_lds_bytes_rows': (c_int64, [c_int, c_int, c_int, c_int]),
    'tfep_inverse_block_lds_bytes_paired': (c_int64, [c_int, c_int, c_int]),
    'tfep_inverse_block': (c_int, [POINTER(InverseBlockDesc), _P]),
    'tfep_diag_inverse_cycles': (c_int, [_P]),
    'tfep_maf_layer_tile_n': (c_int, []),
    'tfep_maf_layer_forward_split': (c_int, [POINTER(MafLayerDesc), _P]),
    'tfep_diag_maf_layer_cycles': (c_int, [_P]),
    'tfep_diag_mfma_peak': (c_int, [_P, c_int, c_int, _P]),
    'tfep_masked_linear_gemm': (c_int, [POINTER(GemmDesc), _P]),
    'tfep_transpose': (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, _P]),
    'tfep_transpose_split': (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, c_int, c_int, _P, _P, _P]),
    'tfep_transpose_split_rows': (c_int, [_P, c_int64, c_int, c_int, _P, c_int64, _P]),
    'tfep_column_sums_absmax': (c_int, [_P, c_int64, c_int, c_int, _P, c_int, _P, _P]),
    'tfep_column_sums': (c_int, [_P, c_int64, c_int, c_int, _P, c_int, _P]),
    'tfep_add_inplace': (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, _P]),
    'tfep_affine_backward': (c_int, [_P, c_int64, _P, ParamLayout, _P, c_int64, _P, _P, ParamLayout, _P, c_int64,
                                     c_int, c_int, _P]),
    'tfep_spline_backward': (c_int, [_P, c_int64, _P, ParamLayout, POINTER(SplineDesc), _P, c_int64, _P, _P,
                                     ParamLayout, _P, c_int64, c_int, c_int, _P]),
    'tfep_moebius_backward': (c_int, [_P, c_int64, _P, c_int64, c_int, c_float, c_int, c_int, _P, c_int64, _P,
                                      _P, c_int64, _P, c_int64, c_int, c_int, _P]),
    'tfep_copy_2d': (c_int, [_P, c_int64, _P, c_int64, c_int, c_int, _P]),
    'tfep_weight_norm_backward': (c_int, [_P, c_int64, _P, _P, _P, c_int, c_int, _P, _P, _P, _P, _P]),
    'tfep_weight_norm_backward_prefix': (c_int, [_P, c_int64, _P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P]),
    'tfep_periodic_embedding_backward': (c_int, [_P, c_int64, _P, c_int, _P, c_int, c_float, c_float, _P, c_int64,
                                                 _P, c_int64, c_int, _P]),
    'tfep_tfep_reduce': (c_int, [_P, _P, _P, _P, _P, c_float, c_int, c_int, _P, _P, _P]),
    'tfep_tfep_reduce_workspace_doubles': (c_int, [c_int]),
    'tfep_bootstrap_fep': (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_int64, c_float, _P, _P]),
    'tfep_egnn_tile': (c_int, [c_int, c_int]),
    'tfep_egnn_packed_floats': (c_int64, [c_int]),
    'tfep_egnn_pack_layer': (c_int, [POINTER(EgnnLayerParams), c_int, _P, _P]),
    'tfep_egnn_embed': (c_int, [_P, c_int, c_int, c_float, _P, _P, c_int, _P, _P, POINTER(EgnnLayerParams), c_int,
                                _P, _P, _P, _P]),
    'tfep_egnn_edge': (c_int, [POINTER(EgnnEdgeArgs), _P]),
    'tfep_egnn_node': (c_int, [POINTER(EgnnNodeArgs), _P]),
    'tfep_egnn_finish': (c_int, [_P, _P, _P, _P, c_int, c_int, _P, _P, c_float, _P, _P, _P, _P]),
    'tfep_egnn_edge_backward': (c_int, [POINTER(EgnnEdgeBwdArgs), _P]),
    'tfep_egnn_node_backward': (c_int, [POINTER(EgnnNodeBwdArgs), _P]),
    'tfep_egnn_center': (c_int, [_P, c_int, c_int, c_float, _P, _P]),
    'tfep_row_dots': (c_int, [_P, _P, c_int, c_int, c_float, _P, _P, _P]),
    'tfep_radial_expansion': (c_int, [_P, c_int64, _P, _P, c_int, c_float, c_int, c_int, _P, _P]),
    'tfep_segment_sum': (c_int, [_P, _P, c_int64, c_int, c_int64, _P, _P]),
    'tfep_ode_axpy': (c_int, [_P, POINTER(c_void_p), POINTER(c_float), c_int, c_int64, _P, _P]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))


class TfepHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once) and bind every declared entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and 'TFEP_HIP_LIB' not in os.environ:
        # Not a fallback: compile the HIP sources in-tree (hipcc, ~10 s) if the library did not travel.
        try:
            from . import build as _build
            _build.build(verbose=False)          # under a file lock, atomic rename: safe with one process per GPU
        except Exception as e:  # noqa: BLE001
            raise TfepHipError(
                f'{LIB_PATH} not found and building it failed ({e}); run `python -m tfep_amd.build` '
                '(tfep_amd has no CPU fallback).') from e
    if not os.path.exists(LIB_PATH):
        raise TfepHipError(
            f'{LIB_PATH} not found: build it with `python -m tfep_amd.build` '
            '(tfep_amd has no CPU fallback).')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    v = lib.tfep_hip_abi_version()
    if v != ABI_VERSION:
        raise TfepHipError(f'libtfep_hip.so ABI version {v} != expected {ABI_VERSION}; rebuild it.')
    _lib = lib
    return lib


def call(name, *args):
    """Call an int-returning entry point; raise with the library's message on failure."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.tfep_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise TfepHipError(f'{name} failed ({rc}): {msg}')


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


def stream_of(t):
    """The current HIP stream of the tensor's device.  Kernels launch on the CURRENT device: a tensor that lives on
    another GPU is an error here (wrap the call in ``torch.cuda.device(tensor.device)``), never a silent cross-device
    launch."""
    cur = torch.cuda.current_device()
    idx = t.device.index
    if idx is not None and idx != cur:
        raise TfepHipError(f'tensor on {t.device} but the current device is cuda:{cur}: '
                           'run under torch.cuda.device(tensor.device)')
    # the raw handle without building a torch.cuda.Stream object: ~5 us less on each of the ~25 launches of a small layer
    return c_void_p(_raw_stream(cur))


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None) or (
    lambda index: torch.cuda.current_stream(index).cuda_stream)


def check_device_tensor(t, name, dtype=torch.float32):
    """The kernels take float32 HIP tensors only; anything else is an error (no fallback)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f'{name} must be a torch.Tensor')
    if not t.is_cuda:
        raise TfepHipError(
            f'{name} is on {t.device}: tfep_amd runs on a HIP device only (there is no CPU fallback).')
    if t.dtype != dtype:
        raise TypeError(f'{name} must be {dtype}, got {t.dtype}')
    return t


def rows(t, name):
    """A 2-D float32 HIP tensor with unit column stride (copied if needed); returns (tensor, row stride)."""
    check_device_tensor(t, name)
    if t.dim() != 2:
        raise ValueError(f'{name} must be 2-D (batch, features), got shape {tuple(t.shape)}')
    B, D = t.shape
    ok = (D <= 1 or t.stride(1) == 1) and (B <= 1 or t.stride(0) >= max(D, 1))
    if not ok:
        t = t.contiguous()
    ld = t.stride(0) if (B > 1 and D > 0) else max(D, 1)
    return t, ld
