"""``tfep_maf_layer_forward_split`` (csrc/maf_layer.hip): one MAF layer -- MADE conditioner + Moebius(d = 2) transformer +
log|det J| -- in one launch, against the reference goldens and against the same arithmetic launched kernel by kernel."""
import ctypes
import math

import numpy as np
import pytest
import torch

import golden_util as gu
from tfep_amd import _lib
from tfep_amd.nn.conditioners import generate_degrees
from tfep_amd.nn.flows import MAF, SequentialFlow
from tfep_amd.nn.transformers import MoebiusTransformer

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()


def _unit_vectors(B, n_vec, seed, radius=None):
    g = torch.Generator('cuda').manual_seed(seed)
    ang = torch.rand(B, n_vec, device='cuda', generator=g) * (2 * math.pi)
    r = 1.0 if radius is None else (0.2 + radius * torch.rand(B, n_vec, device='cuda', generator=g))[..., None]
    return (torch.stack([torch.cos(ang), torch.sin(ang)], dim=-1) * r).reshape(B, 2 * n_vec).contiguous()


def test_reference_golden_flow_through_the_layer_kernel():
    """The 'moeb' golden (2-layer MAF + Moebius(2, unit sphere), 12 features: one partial column tile, a ragged row tile)
    against the reference in float64, tolerances of test_flow_forward."""
    from test_gpu_parity import check_ldj, check_y
    g = gu.load('flows.npz')
    flow = gu.build_flow('moeb', g)
    for layer in flow:
        layer.layer_kernel = True
    x = dev(g['moeb/x'])
    x_before = x.clone()
    with torch.no_grad():
        assert all(layer._layer_kernel_ok(x) for layer in flow)
        y, l = flow(x)
    assert torch.equal(x, x_before), 'input modified'
    check_y(y, g['moeb/y_f64'])
    ref_noise = np.abs(g['moeb/ldj_f32'].astype(np.float64) - g['moeb/ldj_f64']).max()
    check_ldj(l, g['moeb/ldj_f64'], floor=max(4 * ref_noise, 2e-5))


@pytest.mark.parametrize('n_vec,hidden,unit,order,B', [
    (300, 2, True, 'ascending', 1000),        # 600 features: 3 column tiles, the last one partial; 4 row tiles, the last ragged
    (300, 2, False, 'descending', 777),       # general radius: the full Jacobian, rescaled parameters
    (512, 2, True, 'descending', 2048),       # cfg4-ii's layer
    (128, 1, True, 'ascending', 300),         # one hidden layer: one scratch panel
    (200, 3, False, 'ascending', 513),        # three hidden layers: the panels alternate twice
    (70, [160, 352], True, 'ascending', 256),  # hidden layers of different widths
])
def test_layer_kernel_equals_the_launch_by_launch_split_path(n_vec, hidden, unit, order, B):
    D = 2 * n_vec
    torch.manual_seed(n_vec + B)
    with torch.device('cuda'):
        layer = MAF(generate_degrees(D, order, repeats=2), transformer=MoebiusTransformer(dimension=2, unit_sphere=unit),
                    hidden_layers=hidden, initialize_identity=False)
    x = _unit_vectors(B, n_vec, seed=B, radius=None if unit else 1.5)
    layer.split_gemm = True
    with torch.no_grad():
        layer.layer_kernel = True
        assert layer._layer_kernel_ok(x)
        y1, l1 = layer(x)
        layer.layer_kernel = False
        assert not layer._layer_kernel_ok(x)
        y0, l0 = layer(x)
        layer.split_gemm = False                      # the exact-fp32 kernels: the yardstick for both
        ye, le = layer(x)
    assert bool(torch.isfinite(y1).all()) and bool(torch.isfinite(l1).all())
    # the same split arithmetic: dot products in the same order, the same row scales, the same fp64 map
    assert float((y1 - y0).abs().max()) <= 2e-6, float((y1 - y0).abs().max())
    assert float((l1 - l0).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max()))
    # ... and no further from the exact-fp32 path than the launch-by-launch split path is
    err1, err0 = float((y1 - ye).norm() / ye.norm()), float((y0 - ye).norm() / ye.norm())
    assert err1 <= max(2 * err0, 1e-6), (err1, err0)
    assert float((l1 - le).abs().max()) <= max(2 * float((l0 - le).abs().max()), 1e-5 * max(1.0, float(le.abs().max())))
    if unit:
        assert float((y1.reshape(B, n_vec, 2).norm(dim=-1) - 1).abs().max()) < 1e-5


def test_rows_do_not_depend_on_the_batch_they_sit_in():
    n_vec, B = 256, 3000
    D = 2 * n_vec
    torch.manual_seed(3)
    with torch.device('cuda'):
        flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending', repeats=2),
                                    transformer=MoebiusTransformer(dimension=2, unit_sphere=True), initialize_identity=False)
                                for i in range(2)])
    for layer in flow:
        layer.split_gemm, layer.layer_kernel = True, True
    x = _unit_vectors(B, n_vec, seed=11)
    with torch.no_grad():
        y, l = flow(x)
        y2, l2 = flow(x)
        assert torch.equal(y, y2) and torch.equal(l, l2)                      # deterministic
        for lo, hi in ((0, 100), (1234, 1234 + 257), (B - 31, B)):
            ys, ls = flow(x[lo:hi].clone())
            assert torch.equal(ys, y[lo:hi]) and torch.equal(ls, l[lo:hi]), (lo, hi)
        # a view with a row stride (the kernel wants 8-byte aligned rows: an odd offset is copied, an even one is not)
        wide = torch.zeros(64, D + 6, device='cuda')
        wide[:, 2:D + 2] = x[:64]
        yv, lv = flow(wide[:, 2:D + 2])
        assert torch.equal(yv, y[:64]) and torch.equal(lv, l[:64])
        wide[:, 3:D + 3] = x[:64]
        yv, lv = flow(wide[:, 3:D + 3])
        assert torch.equal(yv, y[:64]) and torch.equal(lv, l[:64])
        xi, li = flow.inverse(y[:200])
        assert float((xi - x[:200]).abs().max()) < 2e-4
        assert torch.allclose(li + l[:200], torch.zeros(200, device='cuda'), atol=2e-3)


def test_layer_kernel_is_only_taken_where_it_exists():
    from tfep_amd.nn.transformers import AffineTransformer
    D = 16
    x = torch.randn(8, D, device='cuda')
    with torch.device('cuda'):
        aff = MAF(generate_degrees(D), transformer=AffineTransformer())
        m3 = MAF(generate_degrees(18, repeats=3), transformer=MoebiusTransformer(dimension=3))
        cond = MAF(generate_degrees(D, conditioning_indices=[0, 1], repeats=2), transformer=MoebiusTransformer(dimension=2))
        ok = MAF(generate_degrees(D, repeats=2), transformer=MoebiusTransformer(dimension=2))
    for layer in (aff, m3, cond, ok):
        layer.layer_kernel = True
    assert not aff._layer_kernel_ok(x) and not m3._layer_kernel_ok(torch.randn(8, 18, device='cuda')) and not cond._layer_kernel_ok(x)
    assert ok._layer_kernel_ok(x)
    ok.layer_kernel = None                               # opt-in: measured slower than the launch-by-launch path
    assert not ok._layer_kernel_ok(x)
    with torch.no_grad():
        y, l = cond(x)                                   # the generic path still serves them
    assert bool(torch.isfinite(y).all())


def test_c_abi_rejects_bad_descriptors():
    lib = _lib.load()
    assert lib.tfep_maf_layer_tile_n() in (128, 256)
    assert lib.tfep_maf_layer_forward_split(None, None) != 0 and b'NULL descriptor' in lib.tfep_last_error()
    d = _lib.MafLayerDesc()
    d.B = 0
    assert lib.tfep_maf_layer_forward_split(ctypes.byref(d), None) == 0       # empty batch: nothing to do
    d.B, d.n_linears = 4, 1
    assert lib.tfep_maf_layer_forward_split(ctypes.byref(d), None) != 0 and b'masked linears' in lib.tfep_last_error()
    d.n_linears, d.kind, d.moebius_dim = 3, 1, 2
    assert lib.tfep_maf_layer_forward_split(ctypes.byref(d), None) != 0 and b'Moebius' in lib.tfep_last_error()
    d.kind = 2
    assert lib.tfep_maf_layer_forward_split(ctypes.byref(d), None) != 0 and b'NULL pointer' in lib.tfep_last_error()


def test_split_rows_handed_over_between_unit_sphere_moebius_layers():
    """cfg4-ii's layers on the launch-by-launch split path: the Moebius map writes its output as split-f16 rows too (scale of
    the bound |y| <= 1) and leaves them on the tensor; the next layer's first GEMM takes them instead of a conversion pass
    (``tfep_moebius_forward_split_out``).  Same result as with the pass (the row scale differs by at most one power of two: the
    split format's own rounding), nothing handed over once the tensor has been written or when the next layer embeds it."""
    import os
    n_vec, B = 256, 1500
    D = 2 * n_vec
    torch.manual_seed(5)
    with torch.device('cuda'):
        flow = SequentialFlow(*[MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending', repeats=2),
                                    transformer=MoebiusTransformer(dimension=2, unit_sphere=True), initialize_identity=False)
                                for i in range(3)])
    for layer in flow:
        layer.split_gemm = True
    x = _unit_vectors(B, n_vec, seed=21)
    with torch.no_grad():
        y1, l1 = flow(x)
        mid, _ = flow[0](x)
        assert getattr(mid, '_tfep_split', None) is not None and mid._tfep_split[0].shape == (B, D)
        os.environ['TFEP_SPLIT_HANDOVER'] = '0'
        try:
            y0, l0 = flow(x)
            mid0, _ = flow[0](x)
            assert getattr(mid0, '_tfep_split', None) is None
        finally:
            del os.environ['TFEP_SPLIT_HANDOVER']
        assert torch.equal(mid, mid0)                                  # the fp32 output itself is the same kernel's
        assert float((y1 - y0).abs().max()) <= 2e-6 and float((l1 - l0).abs().max()) <= 2e-5 * max(1.0, float(l0.abs().max()))
        # a tensor written after the fact hands nothing over: the stale rows must not be used
        mid.mul_(1.0)
        ya, la = flow[1](mid)
        yb, lb = flow[1](mid0.clone())
        assert torch.equal(ya, yb) and torch.equal(la, lb)
