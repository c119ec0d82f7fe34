"""GPU edge cases: odd shapes, every hidden-layer depth, large bin counts, non-contiguous inputs, and the
C ABI's error convention (negative status + message, never a throw or a fault)."""
import ctypes

import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import flows as oflows, made as omade

pytestmark = pytest.mark.gpu


def oracle_flow(flow, specs, x):
    sd = {k: (v.cpu().numpy().astype(np.float64) if v.dtype == torch.float32 else v.cpu().numpy())
          for k, v in flow.state_dict().items()}
    layers = [dict(degrees_in=s['degrees_in'], transformer=s['transformer'], embedding=None,
                   made=omade.made_layers_from_state(sd, prefix=f'{i}._conditioner.')) for i, s in enumerate(specs)]
    return oflows.sequential_forward(x.cpu().numpy().astype(np.float64), layers)


@pytest.mark.parametrize('D,B,hidden,K', [(2, 1, 2, 3), (3, 5, 1, 2), (7, 33, 3, 20), (37, 300, 1, 8), (130, 17, 2, 12)])
def test_odd_shapes_depths_and_bin_counts(D, B, hidden, K):
    """D=2 (a single conditioning input; D=1 has no relevant input and fails in the reference too,
    conditioners/made.py:447), B=1, 1-3 hidden layers, K up to 20 (the KMAX=32 kernel),
    feature counts that are not multiples of 4 / 16 -- fused (where supported) and generic paths."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(D * 100 + K)
    orders = ['ascending', 'descending']
    flow = SequentialFlow(*[MAF(generate_degrees(D, o),
                                transformer=NeuralSplineTransformer(torch.full((D,), -3.0), torch.full((D,), 3.0), K),
                                hidden_layers=hidden, initialize_identity=False) for o in orders])
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1)) * 1.5
    specs = [dict(degrees_in=omade.generate_degrees(D, o),
                  transformer=dict(type='spline', x0=np.full(D, -3.0), xf=np.full(D, 3.0), n_bins=K)) for o in orders]
    y_ref, l_ref = oracle_flow(flow, specs, x)
    flow = flow.cuda()
    with torch.no_grad():
        for fused in (True, False):
            for layer in flow:
                layer.fused = fused
            y, l = flow(x.cuda())
            assert gu.err_stats(y.cpu().numpy(), y_ref)[0] < 1e-5
            np.testing.assert_allclose(l.cpu().numpy(), l_ref, rtol=1e-5, atol=5e-5)
        xi, li = flow.inverse(y)
        assert torch.allclose(xi.cpu(), x, atol=2e-4)
        assert torch.allclose(li + l, torch.zeros_like(l), atol=2e-3)


def test_non_contiguous_and_strided_inputs():
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    torch.manual_seed(0)
    maf = MAF(generate_degrees(6), initialize_identity=False).cuda()
    base = torch.randn(6, 40, device='cuda')
    xt = base.t()                                  # (40, 6) with column stride 40
    big = torch.randn(40, 9, device='cuda')
    xs = big[:, 2:8]                               # row stride 9, unit column stride, offset pointer
    with torch.no_grad():
        for x in (xt, xs):
            assert not x.is_contiguous()
            y, l = maf(x)
            y2, l2 = maf(x.contiguous())
            assert torch.equal(y, y2) and torch.equal(l, l2)


def test_c_abi_error_convention():
    """Bad arguments return a negative status and set tfep_last_error(); nothing is launched."""
    from tfep_amd import _lib
    lib = _lib.load()
    x = torch.zeros(4, 8, device='cuda')
    lay = _lib.ParamLayout(16, 8, 1)
    rc = lib.tfep_affine_forward(_lib.ptr(x), 8, None, lay, _lib.ptr(x), 8, None, 0, 4, 8, None)
    assert rc == -1 and b'non-NULL' in lib.tfep_last_error()
    desc = _lib.SplineDesc(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 64, 0, 0, 0, 0, 1e-4, 1e-4)
    rc = lib.tfep_spline_forward(_lib.ptr(x), 8, _lib.ptr(x), lay, ctypes.byref(desc), _lib.ptr(x), 8, None, 0, 4, 8, None)
    assert rc == -1 and b'n_bins=64' in lib.tfep_last_error()
    desc = _lib.SplineDesc(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), 4, 1, 0, 1, 0, 1e-4, 1e-4)
    rc = lib.tfep_spline_forward(_lib.ptr(x), 8, _lib.ptr(x), lay, ctypes.byref(desc), _lib.ptr(x), 8, None, 0, 4, 8, None)
    assert rc == -1 and b'circular spline with learnable limits' in lib.tfep_last_error()
    # GEMM operand contract: k_padded must be a multiple of the k tile, operands 16-byte aligned
    w = torch.zeros(16, 20, device='cuda')
    rc = lib.tfep_masked_linear_forward(_lib.ptr(x), 8, _lib.ptr(w), 20, None, None, None, None, _lib.ptr(x), 8,
                                        4, 8, 16, 20, 0, 0, None)
    assert rc == -1 and b'k_padded' in lib.tfep_last_error()
    rc = lib.tfep_moebius_forward(_lib.ptr(x), 8, _lib.ptr(x), 8, 3, 0.99, 1, 1, _lib.ptr(x), 8, None, 0, 4, 8, None)
    assert rc == -1 and b'multiple of dimension' in lib.tfep_last_error()
    # the Python host turns argument errors into ValueError, like the reference's constructor checks
    from tfep_amd import ops
    with pytest.raises(ValueError, match='must have shape'):
        ops.moebius(x, torch.zeros(4, 7, device='cuda'), 2)


def test_nan_inputs_propagate_without_faults():
    """No runtime error path in forward: NaNs propagate, only to the rows that hold them."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    torch.manual_seed(0)
    maf = MAF(generate_degrees(20), initialize_identity=False).cuda()
    x = torch.randn(300, 20, device='cuda')
    x[7, 3] = float('nan')
    with torch.no_grad():
        y, l = maf(x)
        ok = torch.ones(300, dtype=torch.bool, device='cuda')
        ok[7] = False
        assert torch.isfinite(y[ok]).all() and torch.isfinite(l[ok]).all()
        assert torch.isnan(l[7]) and torch.isnan(y[7]).any()


def test_hip_graph_replay_matches_eager_and_tracks_parameter_updates():
    """GraphedFlow: forward and blocked inverse captured in a HIP graph reproduce the eager results bit for
    bit, also after an in-place parameter update (the weight re-pack is inside the graph)."""
    from tfep_amd.graphs import GraphedFlow
    g = gu.load('flows.npz')
    flow = gu.build_flow('cfg1', g)
    x = torch.from_numpy(g['cfg1/x'][:256]).cuda()
    with torch.no_grad():
        y0, l0 = flow(x)
        gf = GraphedFlow(flow, 256, 66)
        y1, l1 = gf(x)
        assert torch.equal(y0, y1) and torch.equal(l0, l1)
        gi = GraphedFlow(flow, 256, 66, inverse=True)
        xi, li = gi(y0)
        xe, le = flow.inverse(y0)
        assert torch.equal(xi, xe) and torch.equal(li, le)
        for p in flow.parameters():
            p.mul_(1.01)
        y2, l2 = gf(x)
        ye, le2 = flow(x)
        assert torch.equal(y2, ye) and torch.equal(l2, le2) and not torch.equal(y2, y1)
    with pytest.raises(ValueError, match='captured for shape'):
        gf(x[:10])


def test_graft_entry_smoke_runs():
    """The driver's round-end smoke test: forward vs oracle + one backward."""
    import importlib
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module('__graft_entry__')
    entry.smoke()


def test_new_entry_points_reject_bad_arguments():
    """Argument validation of the split-GEMM / inverse-block / bootstrap entry points: an error code and a message,
    never a launch with inconsistent shapes."""
    import ctypes
    from tfep_amd import _lib, ops
    a = torch.randn(8, 64, device='cuda')
    out = torch.empty(8, 64, device='cuda')
    inv = torch.empty(8, device='cuda')
    with pytest.raises(ValueError, match='cols_padded'):
        _lib.call('tfep_split_rows', _lib.ptr(a), 64, 8, 64, _lib.ptr(out), 64, 48, _lib.ptr(inv), 0, _lib.stream_of(a))
    with pytest.raises(ValueError, match='NULL'):
        _lib.call('tfep_split_rows', None, 64, 8, 64, _lib.ptr(out), 64, 64, _lib.ptr(inv), 0, _lib.stream_of(a))
    # split GEMM: k_padded must be a multiple of 32, scales must be given
    d = _lib.GemmDesc()
    w = torch.randn(32, 48, device='cuda')
    d.x, d.ldx, d.w, d.ldw, d.y, d.ldy = a.data_ptr(), 48, w.data_ptr(), 48, out.data_ptr(), 64
    d.B, d.N, d.n_rows_w, d.k_padded, d.split = 8, 32, 32, 48, 1
    d.x_inv_scale, d.w_inv_scale = inv.data_ptr(), inv.data_ptr()
    with pytest.raises(ValueError, match='multiple of 32'):
        _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), _lib.stream_of(a))
    d.k_padded, d.ldx, d.ldw, d.x_inv_scale = 64, 64, 64, None
    with pytest.raises(ValueError, match='scale'):
        _lib.call('tfep_masked_linear_gemm', ctypes.byref(d), _lib.stream_of(a))
    # split-K needs a plain linear product
    d2 = _lib.GemmDesc()
    d2.x, d2.ldx, d2.w, d2.ldw, d2.y, d2.ldy = a.data_ptr(), 64, a.data_ptr(), 64, out.data_ptr(), 64
    d2.B, d2.N, d2.n_rows_w, d2.k_padded, d2.act, d2.k_split, d2.slab_stride = 8, 8, 8, 64, 1, 2, 512
    with pytest.raises(ValueError, match='k_split'):
        _lib.call('tfep_masked_linear_gemm', ctypes.byref(d2), _lib.stream_of(a))
    # inverse block: descriptor checks
    ib = _lib.InverseBlockDesc()
    ib.B, ib.n_layers, ib.n_steps, ib.kind = 8, 5, 1, 0
    with pytest.raises(ValueError, match='hidden layers'):
        _lib.call('tfep_inverse_block', ctypes.byref(ib), _lib.stream_of(a))
    ib.n_layers = 1
    with pytest.raises(ValueError, match='NULL'):
        _lib.call('tfep_inverse_block', ctypes.byref(ib), _lib.stream_of(a))
    # bootstrap: weights with biased data, non-positive kT
    work = torch.randn(16, device='cuda')
    res = torch.empty(2, dtype=torch.float64, device='cuda')
    wts = torch.full((2, 16), 1 / 16, device='cuda')
    with pytest.raises(ValueError, match='not supported with biased'):
        _lib.call('tfep_bootstrap_fep', _lib.ptr(work), _lib.ptr(work), None, _lib.ptr(wts), 16, 2, 16, 1.0, _lib.ptr(res),
                  _lib.stream_of(work))
    with pytest.raises(ValueError, match='kT'):
        _lib.call('tfep_bootstrap_fep', _lib.ptr(work), None, None, _lib.ptr(wts), 16, 2, 16, 0.0, _lib.ptr(res),
                  _lib.stream_of(work))
    assert ops.tile_sizes()[2] == _lib.load().tfep_split_tile_k() == 32
    # reductions / incremental split conversion / inverse-block sizes
    with pytest.raises(ValueError, match='mode must be 0'):
        _lib.call('tfep_abs_reduce', _lib.ptr(a), 64, 8, 64, 2, _lib.ptr(inv), _lib.stream_of(a))
    with pytest.raises(ValueError, match='ld_src < cols'):
        _lib.call('tfep_abs_reduce', _lib.ptr(a), 32, 8, 64, 0, _lib.ptr(inv), _lib.stream_of(a))
    with pytest.raises(ValueError, match='must lie inside both rows'):
        _lib.call('tfep_split_columns_scaled', _lib.ptr(a), 64, 8, 56, 16, _lib.ptr(out), 64, _lib.ptr(inv), _lib.stream_of(a))
    lib = _lib.load()
    assert lib.tfep_inverse_block_lds_bytes(0, 10, 10) == -1 and lib.tfep_inverse_block_lds_bytes(2, -1, 10) == -1
    small, big = lib.tfep_inverse_block_lds_bytes(2, 100, 16), lib.tfep_inverse_block_lds_bytes(2, 400, 64)
    assert 0 < small < 160 * 1024 < big                      # the second one must make the planner shrink its blocks
    ib2 = _lib.InverseBlockDesc()
    ib2.B, ib2.n_layers, ib2.n_steps, ib2.kind = 8, 1, 1, 4
    with pytest.raises(ValueError, match='kind must be 0'):
        _lib.call('tfep_inverse_block', ctypes.byref(ib2), _lib.stream_of(a))


def test_weight_prepack_on_side_stream_is_transparent_and_never_stale():
    """SequentialFlow packs layer i+1's weights on a side stream while layer i computes: same results as without the
    overlap, and weights packed ahead are discarded when a parameter changed in the meantime."""
    import os
    if os.environ.get('TFEP_SPLIT_GEMM', '1') == '0' or os.environ.get('TFEP_OVERLAP_PACK', '1') == '0':
        pytest.skip('needs the default split-GEMM forward with side-stream packing')
    import os
    from tfep_amd.nn.flows.sequential import _side_stream
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D = 300                                             # large enough for the overlap to switch on (>= 4 M weights)
    torch.manual_seed(0)
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -5.), torch.full((D,), 5.), 8),
                                initialize_identity=False) for o in ('ascending', 'descending')]).cuda()
    assert sum(lin.mask.numel() for lin in flow[1]._conditioner._linears()) >= 1 << 22
    x = torch.randn(500, D, device='cuda')
    with torch.no_grad():
        y1, l1 = flow(x)
        os.environ['TFEP_OVERLAP_PACK'] = '0'
        try:
            y0, l0 = flow(x)
        finally:
            del os.environ['TFEP_OVERLAP_PACK']
        assert torch.equal(y0, y1) and torch.equal(l0, l1)
        assert all(layer._conditioner._packed_ahead is None for layer in flow)        # consumed and dropped
        # pack layer 1 ahead, then change its parameters: the stale pack must not be used
        layer = flow[1]
        h = flow[0](x)[0]
        layer.prepack_async(x.device, _side_stream(x.device))
        assert layer._conditioner._packed_ahead is not None
        for p in layer.parameters():
            p.mul_(1.25)
        got = layer(h)
        assert layer._conditioner._packed_ahead is None
        ref = layer(h)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        # and an up-to-date pack IS used, with the same result
        layer.prepack_async(x.device, _side_stream(x.device))
        got2 = layer(h)
        assert torch.equal(got2[0], ref[0]) and torch.equal(got2[1], ref[1])


@pytest.mark.parametrize('split,lookahead', [(False, False), (True, False), (True, True)])
def test_graph_replay_of_the_blocked_inverse_equals_the_eager_inverse(split, lookahead):
    """HIP-graph replay of the blocked inverse (block kernel, split-K block GEMMs, and -- ``split`` -- the split-f16
    output-layer GEMM with its bound-based row scales) gives the bits of the eager call, replay after replay, also for a
    second graph captured later.  (Memset nodes broke exactly this once: ``ops.zeros`` / ``fill_zero_kernel``.)"""
    from tfep_amd.graphs import GraphedFlow
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(3)
    D, B = 150, 333
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8),
                                hidden_layers=[1100, 1300], initialize_identity=False) for o in ('ascending', 'descending')]).cuda()
    for layer in flow:
        layer.split_inverse = split
        if lookahead:                      # side-stream GEMMs of the next block inside the capture (fork / join by events)
            layer.inverse_lookahead, layer.inverse_rows_per_wave = True, 64
    x = torch.randn(B, D, device='cuda') * 1.3
    y, _ = flow(x)                                   # (grad mode on: like a training script that then samples)
    xe, le = flow.inverse(y)                         # (values from the blocked path; only a backward() would pay more)
    xe, le = xe.detach(), le.detach()
    with torch.no_grad():
        for layer in flow:
            assert layer._blocked_plan(y.device)['fused'] is not None
            assert (layer._split_inverse_bound(y.device) is not None) == split
        g1 = GraphedFlow(flow, B, D, inverse=True, warmup=1)
        outs = [g1(y) for _ in range(3)]             # back to back
        g2 = GraphedFlow(flow, B, D, inverse=True, warmup=1)
        outs.append(g2(y))
        outs.append(g1(y))                           # the first graph again, after the second capture
    for xg, lg in outs:
        assert torch.equal(xg, xe) and torch.equal(lg, le)
    assert float((xe - x).abs().max()) < 5e-3


@pytest.mark.parametrize('transformer', ['spline', 'affine'])
def test_checkpoint_from_another_degree_order_loads_into_consistent_plans(transformer):
    """Two MAFs built with different random degree orders; after ``load_state_dict`` the second must compute what the
    first computes -- forward (fused + generic), blocked inverse and pass-per-degree inverse -- also when its plans
    were already built for its own degrees before the load (stale k-ranges / blocked plan)."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    D = 23

    def build(seed):
        torch.manual_seed(seed)
        tr = (lambda: NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8)) \
            if transformer == 'spline' else AffineTransformer
        return SequentialFlow(*[MAF(generate_degrees(D, 'random'), transformer=tr(), initialize_identity=False)
                                for _ in range(2)]).cuda()
    a, b = build(3), build(11)
    x = (torch.randn(70, D, generator=torch.Generator().manual_seed(2)) * 1.3).cuda()
    with torch.no_grad():
        b(x), b.inverse(x)                               # plans of b's OWN degrees exist before the load
        b.load_state_dict(a.state_dict(), strict=True)
        ya, la = a(x)
        xa, lia = a.inverse(ya)
        for fused in (True, False):
            for layer in b:
                layer.fused = fused
            yb, lb = b(x)
            assert torch.equal(ya, yb) and torch.equal(la, lb)
        assert all(layer._blocked_ok() for layer in b)
        xb, lib = b.inverse(ya)
        assert torch.equal(xa, xb) and torch.equal(lia, lib)
        for layer in b:
            layer.blocked_inverse = False
        xr, lir = b.inverse(ya)
        assert torch.allclose(xr, xa, rtol=1e-5, atol=2e-5) and torch.allclose(lir, lia, rtol=1e-5, atol=1e-4)
        assert torch.allclose(xa, x, atol=2e-4)


def test_inverse_under_autograd_is_differentiable_and_agrees_with_the_blocked_inverse():
    """Under grad mode ``inverse`` returns the blocked substitution's values on a node whose backward re-runs the
    differentiable route (the reference's pass per degree, flows/_backward.py); the gradient passes a finite-difference check along a random
    direction of y (parity against the reference's autograd: tests/test_gpu_backward.py::test_inverse_is_differentiable...)."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    torch.manual_seed(0)
    maf = MAF(generate_degrees(6, 'ascending'), initialize_identity=False).cuda()
    y = torch.randn(9, 6, device='cuda', requires_grad=True)
    x, l = maf.inverse(y)
    assert x.requires_grad and torch.isfinite(x).all()
    w = torch.randn(9, 6, device='cuda')
    (gy,) = torch.autograd.grad((w * x).sum() + l.sum(), y)
    with torch.no_grad():
        x2, l2 = maf.inverse(y.detach())
        assert not x2.requires_grad and torch.equal(x2, x.detach()) and torch.equal(l2, l.detach())
        v = torch.randn(9, 6, device='cuda')
        h = 1e-2
        f = lambda yy: float(((w * maf.inverse(yy)[0]).double().sum() + maf.inverse(yy)[1].double().sum()))
        fd = (f(y.detach() + h * v) - f(y.detach() - h * v)) / (2 * h)
    assert abs(fd - float((gy * v).double().sum())) < 2e-3 * max(1.0, abs(fd))


@pytest.mark.parametrize('order', ['ascending', 'descending', 'random'])
@pytest.mark.parametrize('periodic', [False, True])
@pytest.mark.parametrize('rows', [64, 16, (16, 4), (16, 'paired')])
def test_lookahead_inverse_equals_the_in_order_inverse(order, periodic, rows):
    """The blocked inverse overlaps the wide GEMMs of block k + 1 (over the hidden units that were complete before block
    k) with block k's kernel on a side stream; what block k adds follows as one more split-K slab.  Same products, one
    more term in the association: results within a few ulp of the in-order schedule, for every degree order (a random
    order keeps its layer-0 GEMM whole: its new input columns are not one range), stable from call to call (no race
    between the side-stream GEMMs and the block kernel's writes), and a round trip."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(11)
    D, B = 600, 517
    deg = generate_degrees(D, order) if order != 'random' else torch.randperm(D)
    lim = (-4.0, 4.0)
    emb = PeriodicEmbedding(n_features_in=D, limits=list(lim), periodic_indices=list(range(0, D, 3))) if periodic else None
    maf = MAF(degrees_in=deg, transformer=NeuralSplineTransformer(torch.full((D,), lim[0]), torch.full((D,), lim[1]), 8),
              hidden_layers=[1300, 1500], embedding=emb, initialize_identity=False).cuda()
    x = (torch.rand(B, D, device='cuda') * 2 - 1) * 3.9
    with torch.no_grad():
        y, lf = maf(x)
        plan = maf._blocked_plan(y.device)
        assert plan['fused'] is not None and len(plan['blocks']) > 2
        looks = [b['fused']['wide0']['look'] for b in plan['blocks'][1:] if b['fused']['wide0'] is not None]
        # (the first blocks know too few columns for a split to pay; the embedding keeps periodic and plain columns apart)
        assert any(looks) or order == 'random' or periodic
        assert maf.inverse_lookahead is None                     # default: by batch, layer size and row layout; forced here
        maf.inverse_lookahead = True
        # (the default pairs look-ahead with 64-row waves, or with 16-row waves packed four to a workgroup: the packed
        # launch -- independent waves, each with its own LDS region -- must give the bits of the unpacked one)
        # ... and a PAIRED launch (a loader wave beside every chain wave, double-buffered stage, one workgroup barrier per
        # stage) the bits of the single-wave one: the consumer does the same arithmetic in the same order
        rows, wpw = rows if isinstance(rows, tuple) else (rows, None)
        paired = wpw == 'paired'
        wpw = None if paired else wpw
        maf.inverse_rows_per_wave, maf.inverse_waves_per_workgroup, maf.inverse_paired = rows, wpw, paired
        x1, l1 = maf.inverse(y)
        if wpw or paired:
            maf.inverse_waves_per_workgroup, maf.inverse_paired = None, False
            xu, lu = maf.inverse(y)
            assert torch.equal(xu, x1) and torch.equal(lu, l1)
            maf.inverse_waves_per_workgroup, maf.inverse_paired = wpw, paired
        x2, l2 = maf.inverse(y)
        maf.inverse_lookahead = False
        x0, l0 = maf.inverse(y)
    assert torch.equal(x1, x2) and torch.equal(l1, l2)
    assert float((x1 - x0).abs().max()) < 2e-5 and float((l1 - l0).abs().max()) < 2e-4
    # (600 sequential degrees of a random-init network amplify rounding in a few rows: the typical row, not the worst)
    assert float((x1 - x).abs().max(dim=1).values.median()) < 1e-3 and float((l1 + lf).abs().median()) < 1e-2


def test_a_tensor_of_another_width_never_reaches_the_kernels():
    """The kernels index x by the layer's feature tables; the reference fails in its first F.linear (RuntimeError)."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    maf = MAF(degrees_in=generate_degrees(12, 'ascending'), initialize_identity=False).cuda()
    for bad in (torch.zeros(5, 11, device='cuda'), torch.zeros(5, 16, device='cuda')):
        with pytest.raises(RuntimeError, match='features'):
            maf(bad)
        with pytest.raises(RuntimeError, match='features'):
            maf.inverse(bad)
    y, ldj = maf(torch.zeros(5, 12, device='cuda'))
    assert y.shape == (5, 12) and ldj.shape == (5,)


@pytest.mark.parametrize('kind', ['spline', 'affine', 'moebius', 'spline+periodic+fixed'])
def test_block_kernel_row_layouts_agree(kind):
    """``tfep_inverse_block`` with one sample row per lane (rows_per_wave 64) and with 16 rows per wave (every dot on
    16 x 16 MFMA tiles, four lanes per row in the transformer inverse): the same inverse to rounding, for a batch that is
    not a multiple of 16 (dead rows store nothing), each reproducible from call to call."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import AffineTransformer, MoebiusTransformer, NeuralSplineTransformer
    torch.manual_seed(21)
    D, B = 96, 203
    emb, deg = None, generate_degrees(D, 'ascending')
    if kind == 'affine':
        tr = AffineTransformer()
    elif kind == 'moebius':
        deg = generate_degrees(D, 'descending', repeats=3)
        tr = MoebiusTransformer(dimension=3)
    elif kind == 'spline':
        tr = NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8)
    else:
        deg = generate_degrees(D, 'ascending', conditioning_indices=[5, 50])
        emb = PeriodicEmbedding(D, limits=[-4.0, 4.0], periodic_indices=list(range(2, D, 5)))
        tr = NeuralSplineTransformer(torch.full((D - 2,), -4.0), torch.full((D - 2,), 4.0), 5, circular=False,
                                     identity_boundary_slopes=True)
    maf = MAF(deg, transformer=tr, embedding=emb, hidden_layers=[260, 300], initialize_identity=False).cuda()
    x = (torch.rand(B, D, device='cuda') * 2 - 1) * 3.5
    out = {}
    with torch.no_grad():
        y, lf = maf(x)
        assert maf._blocked_plan(y.device)['fused'] is not None
        for rows in (16, 64):
            maf.inverse_rows_per_wave = rows
            out[rows] = maf.inverse(y)
            again = maf.inverse(y)
            assert torch.equal(out[rows][0], again[0]) and torch.equal(out[rows][1], again[1])
        # the 16-row layout with and without the loader wave (every kind of the block kernel): the same bits
        maf.inverse_rows_per_wave = 16
        for paired in (True, False):
            maf.inverse_paired = paired
            xp, lp = maf.inverse(y)
            assert torch.equal(xp, out[16][0]) and torch.equal(lp, out[16][1]), paired
        maf.inverse_paired = None
    assert float((out[16][0] - out[64][0]).abs().max()) < 5e-5 and float((out[16][1] - out[64][1]).abs().max()) < 5e-4
    assert float((out[16][0] - x).abs().max(dim=1).values.median()) < 1e-3
    # the entry point rejects any other layout
    from tfep_amd import _lib
    d = _lib.InverseBlockDesc()
    d.B, d.n_steps, d.n_layers, d.rows_per_wave = 4, 1, 1, 32
    assert _lib.load().tfep_inverse_block(ctypes.byref(d), None) != 0
    assert _lib.load().tfep_inverse_block_lds_bytes_rows(2, 100, 16, 32) == -1
    assert 0 < _lib.load().tfep_inverse_block_lds_bytes_rows(2, 100, 16, 16) < _lib.load().tfep_inverse_block_lds_bytes_rows(2, 100, 16, 64)


@pytest.mark.parametrize('kind', ['affine', 'spline+periodic', 'spline, activations kept when eager'])
def test_graphed_training_step_equals_eager_steps(kind):
    """``GraphedTrainingStep``: forward, loss, backward and the optimiser update of a small flow captured into one HIP
    graph.  Replays on a sequence of batches give the losses and the parameters of the same steps run eagerly, bit for
    bit (the eager arm recomputes its activations like the captured one), and keep doing so after the parameters were
    changed from outside (the graph reads the current values)."""
    import copy
    from tfep_amd.graphs import GraphedTrainingStep
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF, SequentialFlow, _backward as bw
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(17)
    D, B = (66, 1024) if kind.startswith('spline,') else (30, 260)     # (6.7 MB of spline parameters: the eager forward keeps them)
    layers = []
    for o in ('ascending', 'descending'):
        if kind == 'affine':
            layers.append(MAF(generate_degrees(D, o), initialize_identity=False))
        elif kind.startswith('spline,'):
            layers.append(MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8),
                              initialize_identity=False))
        else:
            layers.append(MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 6),
                              embedding=PeriodicEmbedding(D, limits=[-4.0, 4.0], periodic_indices=[1, 7, 20]),
                              initialize_identity=False))
    flow = SequentialFlow(*layers).cuda()
    twin = copy.deepcopy(flow)
    xs = [(torch.rand(B, D, device='cuda') * 2 - 1) * 3.5 for _ in range(6)]
    loss_mod = BoltzmannKLDivLoss()
    c = torch.linspace(0.1, 0.5, D, device='cuda')

    def loss_fn(y, ldj):
        return loss_mod((c * y ** 2).sum(dim=1), ldj)
    opt, opt_twin = torch.optim.SGD(flow.parameters(), lr=1e-3, momentum=0.9), torch.optim.SGD(twin.parameters(), lr=1e-3, momentum=0.9)
    step = GraphedTrainingStep(twin, loss_fn, opt_twin, B, D)
    # construction leaves the model and the optimiser as it found them (the warm-up steps are undone in place)
    for (n, p), q in zip(flow.named_parameters(), twin.parameters()):
        assert torch.equal(p, q), n
    assert all(float(v.abs().sum()) == 0.0 for st in opt_twin.state.values() for v in st.values() if torch.is_tensor(v))
    save = bw._SAVE_BYTES
    bw._SAVE_BYTES = 0                                  # (inside a capture the layers recompute their activations)
    try:
        for i, x in enumerate(xs):
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(*flow(x))
            loss.backward()
            opt.step()
            got = step(x)
            assert float(got) == float(loss.detach()), i
            if i == 2:                                  # parameters changed from outside, both arms alike
                with torch.no_grad():
                    for p, q in zip(flow.parameters(), twin.parameters()):
                        p.mul_(1.01)
                        q.mul_(1.01)
    finally:
        bw._SAVE_BYTES = save
    for (n, p), q in zip(flow.named_parameters(), twin.parameters()):
        assert torch.equal(p, q), n
    with pytest.raises(ValueError):
        step(xs[0][:10])
    # an autograd graph over the parameters that is still alive (its gradient accumulators belong to the stream it was built
    # on; a capture that meets them dies inside the HIP runtime): refused up front
    del loss                                                          # (the last eager loss is such a tensor too)
    held = flow(xs[0])
    assert held[0].grad_fn is not None
    with pytest.raises(RuntimeError, match='autograd graph'):
        GraphedTrainingStep(flow, loss_fn, opt, B, D)
    del held
    again = GraphedTrainingStep(flow, loss_fn, opt, B, D)              # gone: captures
    assert bool(torch.isfinite(again(xs[1])))


@pytest.mark.parametrize('kind', ['affine', 'spline', 'moebius'])
def test_empty_batch_through_every_path(kind):
    """B = 0 (the reference returns empty tensors: torch ops on empty batches): forward (fused and generic), blocked
    inverse, the training step with its gradients (zeros), the estimator-side reductions -- nothing launches on an empty
    grid, nothing faults."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import MoebiusTransformer, NeuralSplineTransformer
    D = 12
    tr = {'affine': lambda: None, 'spline': lambda: NeuralSplineTransformer(torch.full((D,), -3.0), torch.full((D,), 3.0), 4),
          'moebius': lambda: MoebiusTransformer(dimension=3)}[kind]
    deg = (lambda o: generate_degrees(D, o, repeats=3)) if kind == 'moebius' else (lambda o: generate_degrees(D, o))
    flow = SequentialFlow(*[MAF(deg(o), transformer=tr(), initialize_identity=False) for o in ('ascending', 'descending')]).cuda()
    x = torch.empty(0, D, device='cuda')
    with torch.no_grad():
        for fused in (True, False):
            for layer in flow:
                layer.fused = fused
            y, l = flow(x)
            assert y.shape == (0, D) and l.shape == (0,)
        xi, li = flow.inverse(y)
        assert xi.shape == (0, D) and li.shape == (0,)
    y, l = flow(x)                                        # under autograd
    (y.sum() + l.sum()).backward()
    for n, p in flow.named_parameters():
        assert p.grad is not None and p.grad.shape == p.shape and float(p.grad.abs().sum()) == 0.0, n
    loss = BoltzmannKLDivLoss()
    with pytest.raises(Exception):
        float(loss(l.detach(), l.detach()))              # the mean over no samples is not a number the reference returns either


def test_graphed_training_step_with_adamw():
    """The optimiser of the reference's maps is AdamW (app/base.py: configure_optimizers): with ``capturable=True`` its update
    is part of the captured step; replays track the same steps run eagerly."""
    import copy
    from tfep_amd.graphs import GraphedTrainingStep
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow, _backward as bw
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(3)
    D, B = 24, 200
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 5),
                                initialize_identity=False) for o in ('ascending', 'descending')]).cuda()
    twin = copy.deepcopy(flow)
    xs = [(torch.rand(B, D, device='cuda') * 2 - 1) * 3.5 for _ in range(5)]
    loss_mod = BoltzmannKLDivLoss()
    c = torch.linspace(0.1, 0.5, D, device='cuda')

    def loss_fn(y, ldj):
        return loss_mod((c * y ** 2).sum(dim=1), ldj)
    kw = dict(lr=1e-3, weight_decay=0.01, capturable=True)
    opt, opt_twin = torch.optim.AdamW(flow.parameters(), **kw), torch.optim.AdamW(twin.parameters(), **kw)
    step = GraphedTrainingStep(twin, loss_fn, opt_twin, B, D, sample_input=xs[0])
    for (n, p), q in zip(flow.named_parameters(), twin.parameters()):          # the warm-up steps were undone in place
        assert torch.equal(p, q), n
    assert all(float(v.abs().sum()) == 0.0 for st in opt_twin.state.values() for v in st.values() if torch.is_tensor(v))
    save = bw._SAVE_BYTES
    bw._SAVE_BYTES = 0
    try:
        for i, x in enumerate(xs):
            opt.zero_grad(set_to_none=True)
            loss = loss_fn(*flow(x))
            loss.backward()
            opt.step()
            got = step(x)
            assert abs(float(got) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach())), i
            del loss
    finally:
        bw._SAVE_BYTES = save
    for (n, p), q in zip(flow.named_parameters(), twin.parameters()):
        assert torch.allclose(p, q, rtol=1e-5, atol=1e-7), n


def test_cached_packed_weights_notice_updates_through_data():
    """``MADE.cache_packed_weights``: served from the cache while nothing changed (bit-identical), re-packed after an
    in-place update -- one that bumps ``Tensor._version`` (``p.mul_`` under no_grad, what optimisers do) and one that does
    not (a write through ``p.data``, which the version counter cannot see: the strided checksum does)."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(4)
    D = 40
    layer = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8),
                initialize_identity=False).cuda()
    x = torch.randn(300, D, device='cuda').clamp_(-3.9, 3.9)
    made = layer._conditioner

    def fresh():
        made.cache_packed_weights = False
        made.invalidate_plan()
        with torch.no_grad():
            out = layer(x)
        made.cache_packed_weights = True
        return out
    for split in (False, True):
        layer.split_gemm = split
        y0, l0 = fresh()
        with torch.no_grad():
            y1, l1 = layer(x)                       # fills the cache
            y2, l2 = layer(x)                       # served from it
        assert torch.equal(y0, y1) and torch.equal(y1, y2) and torch.equal(l1, l2)
        for how in ('versioned', 'data'):
            with torch.no_grad():
                for p in made.parameters():
                    if how == 'versioned':
                        p.mul_(1.01)
                    else:
                        v = p._version
                        p.data.mul_(0.97)
                        assert p._version == v       # (the hazard: nothing in the version says the values changed)
                y3, l3 = layer(x)
            y4, l4 = fresh()
            assert torch.equal(y3, y4) and torch.equal(l3, l4), (split, how)
            assert not torch.equal(y3, y1)
            y1 = y3
    # the blocked inverse keeps its packs across calls under the same switch (a sampling loop packs once) and notices the
    # same two kinds of update
    layer.split_gemm = None
    yy = torch.randn(300, D, device='cuda').clamp_(-3.9, 3.9)

    def fresh_inverse():
        made.cache_packed_weights = False
        made.invalidate_plan()
        with torch.no_grad():
            out = layer.inverse(yy)
        made.cache_packed_weights = True
        return out
    x0, m0 = fresh_inverse()
    with torch.no_grad():
        x1, m1 = layer.inverse(yy)
        n_packed = sum(1 for plan in made._plans.values() for k in plan if isinstance(k, tuple) and k[0] in ('packed', 'packed_split'))
        x2, m2 = layer.inverse(yy)
    assert n_packed > 0                                 # (kept after the call)
    assert torch.equal(x0, x1) and torch.equal(x1, x2) and torch.equal(m1, m2)
    for how in ('versioned', 'data'):
        with torch.no_grad():
            for p in made.parameters():
                if how == 'versioned':
                    p.mul_(1.01)
                else:
                    p.data.mul_(0.97)
            x3, m3 = layer.inverse(yy)
        x4, m4 = fresh_inverse()
        assert torch.equal(x3, x4) and torch.equal(m3, m4), how
        assert not torch.equal(x3, x1)
        x1 = x3
    made.cache_packed_weights = False


def test_large_blocked_inverse_runs_shard_by_shard_with_the_same_results():
    """Above 16 384 rows a long-chain layer is inverted shard by shard on the one-launch-per-super-block schedule, the packed weights
    shared by the shards: row for row the results of calling ``inverse`` on each shard, bit for bit; the packs do not outlive the
    call and ``cache_packed_weights`` is left as it was."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(3)
    D, B, shard = 1100, 16384 + 700, 8192
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8),
              hidden_layers=[1200, 1200], initialize_identity=False).cuda()
    maf.split_inverse = True
    y = (torch.randn(B, D, generator=torch.Generator().manual_seed(9)) * 1.2).cuda()
    made = maf._conditioner
    assert made.cache_packed_weights is False
    with torch.no_grad():
        x, l = maf.inverse(y)
        assert maf.last_inverse_schedule.startswith('sharded')
        assert made.cache_packed_weights is False
        assert not any(isinstance(k, tuple) and k[0] in ('packed', 'packed_split') for plan in made._plans.values() for k in plan)
        for r0 in range(0, B, shard):
            xs, ls = maf.inverse(y[r0:r0 + shard])
            assert torch.equal(x[r0:r0 + shard], xs) and torch.equal(l[r0:r0 + shard], ls)
        maf.inverse_shard_rows = 0
        xw, lw = maf.inverse(y)                              # the whole batch at once: the same map
        assert not maf.last_inverse_schedule.startswith('sharded')
        yy, ll = maf(x)
    assert float((xw - x).abs().max()) < 5e-5 and torch.allclose(lw, l, rtol=1e-5, atol=5e-4)
    assert float((yy - y).abs().max()) < 5e-3 and torch.allclose(ll + l, torch.zeros_like(l), atol=5e-3)


@pytest.mark.parametrize('order,hidden,periodic,B', [('ascending', [1100, 1300], False, 333), ('descending', [900], False, 16),
                                                     ('random', [700, 800, 900], True, 200), ('ascending', [1200, 1200], True, 1)])
def test_super_block_launch_matches_block_by_block_launches(order, hidden, periodic, B):
    """``tfep_inverse_block`` with ``n_blocks`` (one launch per super-block: the pair of waves that owns 16 sample rows forms
    the short products over what the super-block has produced so far itself, exact-fp32 MFMA on its own rows) against the
    block-by-block launches with their short split-f16 GEMMs in between: the same sums in another association -- and against
    the forward map.  Monotone and random degree orders (layer 0 gathers its input columns), a periodic embedding (two
    input entries per feature), one to three hidden layers, a single row and a ragged last wave."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D = 150
    torch.manual_seed(11)
    lo, hi = (0.0, 2.0) if periodic else (-4.0, 4.0)
    emb = PeriodicEmbedding(D, limits=[lo, hi], periodic_indices=list(range(0, D, 3))) if periodic else None
    maf = MAF(generate_degrees(D, order), transformer=NeuralSplineTransformer(torch.full((D,), lo), torch.full((D,), hi), 8, circular=periodic),
              embedding=emb, hidden_layers=hidden, initialize_identity=False).cuda()
    maf.split_inverse = True
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(B, D, generator=gen) * 2.0 if periodic else torch.randn(B, D, generator=gen) * 1.3).cuda()
    with torch.no_grad():
        y, l = maf(x)
        xs, ls = maf.inverse(y)
        assert maf.last_inverse_schedule == 'super_kernel'
        xs2, ls2 = maf.inverse(y)
        assert torch.equal(xs, xs2) and torch.equal(ls, ls2)
        maf.inverse_super_kernel = False
        xb, lb = maf.inverse(y)
        assert maf.last_inverse_schedule == 'block_by_block'
    d, dx = (xs - xb).abs(), (xs - x).abs()
    if periodic:
        d, dx = torch.minimum(d, 2.0 - d), torch.minimum(dx, 2.0 - dx)
    assert float(d.max()) < 5e-5 and torch.allclose(ls, lb, rtol=1e-5, atol=5e-4)
    assert float(dx.max()) < 5e-3 and torch.allclose(ls + l, torch.zeros_like(l), atol=5e-3)
