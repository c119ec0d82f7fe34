"""GPU parity tests: the HIP path (through the C ABI) against the reference's golden vectors
and the CPU oracle.  Run with ``-m gpu`` on an MI355X.

Tolerances (north star: <= 1e-5 relative on mapped coordinates and log|det J|):
  * given identical transformer parameters: relative L2 error of y <= 1e-5 and per-sample
    |d ldj| <= 1e-5 * max(1, |ldj|) against the reference run in float64;
  * end to end through the fp32 GEMMs: relative L2 of y <= 1e-5 and the log-det error no worse
    than 4x the reference's own float32-vs-float64 error on the same data (floor 2e-5 absolute).
"""
import json
import os

import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

REL = 1e-5


def dev(a, dtype=torch.float32):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype).cuda()


def check_y(got, ref, rel=REL, what='y'):
    r, m = gu.err_stats(got.detach().cpu().numpy(), ref)
    assert r <= rel, f'{what}: relative L2 error {r:.3e} > {rel:.1e} (max abs {m:.3e})'
    return r, m


def check_ldj(got, ref, rel=REL, floor=0.0):
    got = got.detach().cpu().numpy().astype(np.float64)
    tol = np.maximum(rel * np.maximum(1.0, np.abs(ref)), floor)
    bad = np.abs(got - ref) > tol
    assert not bad.any(), f'ldj: max abs error {np.abs(got - ref).max():.3e}, {bad.sum()} samples out of tolerance'


# ------------------------------------------------------------------ transformers (identical parameters)

def test_affine_volpres():
    from tfep_amd.nn.transformers import AffineTransformer, VolumePreservingShiftTransformer
    g = gu.load('transformers.npz')
    t = AffineTransformer()
    y, l = t(dev(g['affine/x']), dev(g['affine/par']))
    check_y(y, g['affine/y_f64']); check_ldj(l, g['affine/ldj_f64'])
    x, l = t.inverse(dev(g['affine/inv_in']), dev(g['affine/par']))
    check_y(x, g['affine/xinv_f64'], what='x'); check_ldj(l, g['affine/ldjinv_f64'])

    t = VolumePreservingShiftTransformer(torch.from_numpy(g['volpres/periodic_indices']),
                                         torch.from_numpy(g['volpres/periodic_limits']))
    y, l = t(dev(g['volpres/x']), dev(g['volpres/par']))
    # float32 wrap of a float32 sum: compare against the reference's own float32 result exactly-ish
    np.testing.assert_allclose(y.cpu().numpy(), g['volpres/y_f32'], rtol=0, atol=1e-6)
    assert torch.all(l == 0)
    x, li = t.inverse(dev(g['volpres/y_f32']), dev(g['volpres/par']))
    assert x.shape == y.shape and torch.all(li == 0)
    # value check of the inverse: x = (y - b) wrapped into the period on the periodic columns (affine.py:414-456), in
    # float64 from the same float32 inputs; on the circle a value within rounding of a period boundary may land on
    # either end, so periodic columns are compared modulo the period
    xr = g['volpres/y_f32'].astype(np.float64) - g['volpres/par'].astype(np.float64)
    pidx = g['volpres/periodic_indices'].astype(np.int64)
    lo, hi = (float(v) for v in g['volpres/periodic_limits'])
    xr[:, pidx] = np.mod(xr[:, pidx], hi - lo) + lo
    d = np.abs(x.cpu().numpy().astype(np.float64) - xr)
    d[:, pidx] = np.minimum(d[:, pidx], (hi - lo) - d[:, pidx])
    assert d.max() < 2e-6
    if 'volpres/xinv_f64' in g.files:                      # the reference's own float64 inverse, when the fixture has it
        d = np.abs(x.cpu().numpy().astype(np.float64) - g['volpres/xinv_f64'])
        d[:, pidx] = np.minimum(d[:, pidx], (hi - lo) - d[:, pidx])
        assert d.max() < 2e-6
    # and forward(inverse(y)) == y on the circle
    y2, _ = t(x, dev(g['volpres/par']))
    d = (y2 - dev(g['volpres/y_f32'])).abs().cpu().numpy()
    d[:, pidx] = np.minimum(d[:, pidx], (hi - lo) - d[:, pidx])
    assert d.max() < 2e-6


def _spline_names():
    return sorted(json.loads(str(gu.load('transformers.npz')['spline/meta'])).keys())


@pytest.mark.parametrize('name', _spline_names())
def test_spline_variants(name):
    g = gu.load('transformers.npz')
    meta = json.loads(str(g['spline/meta']))[name]
    t = gu.build_transformer(dict(type='spline', x0=meta['x0'], xf=meta['xf'], n_bins=meta['n_bins'],
                                  y0=meta['y0'], yf=meta['yf'], circular=meta['circular'],
                                  identity_boundary_slopes=meta['identity_boundary_slopes'],
                                  learn_lower_bound=meta['learn_lower_bound'],
                                  learn_upper_bound=meta['learn_upper_bound'])).cuda()
    par = dev(g[name + '/par'])
    y, l = t(dev(g[name + '/x']), par)
    check_y(y, g[name + '/y_f64']); check_ldj(l, g[name + '/ldj_f64'])
    x, l = t.inverse(dev(g[name + '/inv_in']), par)
    check_y(x, g[name + '/xinv_f64'], what='x'); check_ldj(l, g[name + '/ldjinv_f64'])


@pytest.mark.parametrize('name', ['moebius/d2_u0', 'moebius/d2_u1', 'moebius/d3_u0', 'moebius/d3_u1'])
def test_moebius(name):
    from tfep_amd.nn.transformers import MoebiusTransformer
    g = gu.load('transformers.npz')
    meta = json.loads(str(g['moebius/meta']))[name]
    t = MoebiusTransformer(meta['dimension'], meta['max_radius'], meta['unit_sphere'])
    par = dev(g[name + '/par'])
    y, l = t(dev(g[name + '/x']), par)
    check_y(y, g[name + '/y_f64']); check_ldj(l, g[name + '/ldj_f64'])
    x, l = t.inverse(dev(g[name + '/inv_in']), par)
    check_y(x, g[name + '/xinv_f64'], what='x'); check_ldj(l, g[name + '/ldjinv_f64'])


def test_mixed_and_periodic_embedding():
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    g = gu.load('transformers.npz')
    spec = dict(type='mixed',
                transformers=[dict(type='spline', x0=np.full(3, -1.0), xf=np.full(3, 1.0), n_bins=4),
                              dict(type='affine'),
                              dict(type='spline', x0=np.full(2, -1.0), xf=np.full(2, 1.0), n_bins=3, circular=True)],
                indices=[[0, 2, 5], [1, 3], [4, 6]])
    t = gu.build_transformer(spec)
    assert t._parameters_split_indices.tolist() == g['mixed/split'].tolist()
    assert t.get_degrees_out(torch.arange(7)).tolist() == g['mixed/degrees_out'].tolist()
    t = t.cuda()
    par = dev(g['mixed/par'])
    y, l = t(dev(g['mixed/x']), par)
    check_y(y, g['mixed/y_f64']); check_ldj(l, g['mixed/ldj_f64'])
    x, l = t.inverse(dev(g['mixed/inv_in']), par)
    check_y(x, g['mixed/xinv_f64'], what='x'); check_ldj(l, g['mixed/ldjinv_f64'])

    emb = PeriodicEmbedding(6, [0.0, 1.0], periodic_indices=[1, 2, 5])
    assert emb.get_degrees_out(torch.arange(6)).tolist() == g['pemb/degrees_out'].tolist()
    out = emb.cuda()(dev(g['pemb/x']))
    np.testing.assert_allclose(out.cpu().numpy(), g['pemb/y_f64'], rtol=0, atol=2e-6)


def test_empty_and_ragged_batches():
    from tfep_amd.nn.transformers import AffineTransformer
    t = AffineTransformer()
    y, l = t(torch.empty(0, 5, device='cuda'), torch.empty(0, 10, device='cuda'))
    assert y.shape == (0, 5) and l.shape == (0,)
    for B in (1, 3, 5, 257):                  # not multiples of the 4-rows-per-block launch shape
        x = torch.randn(B, 7, device='cuda')
        p = torch.randn(B, 14, device='cuda') * 0.3
        y, l = t(x, p)
        ref = x.double() * torch.exp(p[:, 7:].double()) + p[:, :7].double()
        assert torch.allclose(y.double(), ref, rtol=1e-6, atol=1e-6)
        assert torch.allclose(l.double(), p[:, 7:].double().sum(1), rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError, match='must have shape'):
        t(torch.randn(4, 7, device='cuda'), torch.randn(4, 13, device='cuda'))
    with pytest.raises(TypeError):
        t(torch.randn(4, 7, device='cuda', dtype=torch.float64), torch.randn(4, 14, device='cuda'))


# ------------------------------------------------------------------ masked linear / MADE

def test_masked_linear_and_weight_norm():
    from tfep_amd.nn import masked
    g = gu.load('masked_linear.npz')
    x, w, b, m = dev(g['x']), dev(g['weight']), dev(g['bias']), dev(g['mask'])
    check_y(masked.masked_linear(x, w, b, m), g['y_f64'], rel=2e-6)
    check_y(masked.masked_linear(x, w, b, None), g['y_nomask_f64'], rel=2e-6)
    lin = masked.masked_weight_norm(masked.MaskedLinear(8, 5, mask=torch.from_numpy(g['mask'])))
    lin.load_state_dict({'bias': torch.from_numpy(g['wn_bias']), 'weight_g': torch.from_numpy(g['wn_g']),
                         'weight_v': torch.from_numpy(g['wn_v']), 'mask': torch.from_numpy(g['mask'])})
    lin = lin.cuda()
    weff = lin.weight
    assert torch.all(torch.isfinite(weff)) and torch.all(weff[2] == 0)       # fully-masked row: 0, not NaN
    check_y(weff, g['wn_weight_f64'], rel=1e-6, what='W')
    check_y(lin(x), g['wn_y_f64'], rel=2e-6)
    # extra leading dimensions, like F.linear
    y3 = lin(x.reshape(2, 3, 8))
    assert y3.shape == (2, 3, 5)


@pytest.mark.parametrize('name', ['a', 'b', 'c', 'd'])
def test_made_forward(name):
    from tfep_amd.nn.conditioners import MADE
    g = gu.load('made.npz')
    meta = json.loads(str(g['meta']))[name]
    made = MADE(torch.tensor(meta['degrees_in']), torch.tensor(meta['degrees_out']), meta['hidden_layers'],
                meta['weight_norm'])
    sd = {k: torch.from_numpy(v) for k, v in gu.sub(g, f'{name}/sd/').items()}
    made.load_state_dict(sd, strict=True)
    assert int(made.n_parameters()) == meta['n_parameters']
    made = made.cuda()
    y = made(dev(g[f'{name}/x']))
    check_y(y, g[f'{name}/y_f64'], rel=2e-6, what='params')


# ------------------------------------------------------------------ flows end to end

def _flow_check(name, fused, split=None):
    g = gu.load('flows.npz')
    flow = gu.build_flow(name, g)
    for layer in flow:
        layer.fused = fused
        layer.split_gemm = split
    x = dev(g[f'{name}/x'])
    x_before = x.clone()
    y, l = flow(x)
    assert torch.equal(x, x_before), 'input modified'
    assert y.shape == x.shape and l.shape == (x.shape[0],)
    ry, my = check_y(y, g[f'{name}/y_f64'])
    ref_noise = np.abs(g[f'{name}/ldj_f32'].astype(np.float64) - g[f'{name}/ldj_f64']).max()
    check_ldj(l, g[f'{name}/ldj_f64'], floor=max(4 * ref_noise, 2e-5))
    return flow, g


@pytest.mark.parametrize('split', [None, True])             # None: size heuristic (exact fp32 here); True: split-f16 GEMMs
@pytest.mark.parametrize('fused', [True, False])
@pytest.mark.parametrize('name', ['cfg1', 'rq4', 'cond', 'circ', 'moeb', 'mixflow'])
def test_flow_forward(name, fused, split):
    _flow_check(name, fused, split)


def test_fused_path_is_taken_where_expected():
    g = gu.load('flows.npz')
    assert gu.build_flow('cfg1', g)[0]._fused_kind() == 0
    assert gu.build_flow('rq4', g)[0]._fused_kind() == 1
    assert gu.build_flow('circ', g)[0]._fused_kind() == 1
    assert gu.build_flow('cond', g)[0]._fused_kind() == 0
    assert gu.build_flow('cond', g)[1]._fused_kind() == 1         # identity slopes, K=5: the 14-parameter layout
    assert gu.build_flow('moeb', g)[0]._fused_kind() is None


@pytest.mark.parametrize('name', ['cond', 'circ', 'moeb', 'mixflow', 'rq4'])
def test_flow_inverse(name):
    g = gu.load('flows.npz')
    flow = gu.build_flow(name, g)
    yin = dev(g[f'{name}/inv_in'])
    x, l = flow.inverse(yin)
    rel = 1e-5 if name != 'rq4' else 5e-5       # 66 sequential passes x 4 layers amplify fp32 GEMM noise
    check_y(x, g[f'{name}/xinv_f64'], rel=rel, what='x')
    ref_noise = np.abs(g[f'{name}/ldj_f32'].astype(np.float64) - g[f'{name}/ldj_f64']).max()
    check_ldj(l, g[f'{name}/ldjinv_f64'], floor=max(8 * ref_noise, 5e-5))
    # round trip: forward(inverse(y)) == y and the log-dets cancel (tests/nn/flows/test_maf.py:283-285)
    y2, l2 = flow(x)
    assert torch.allclose(y2, yin, atol=1e-4)
    assert torch.allclose(l + l2, torch.zeros_like(l), atol=1e-3)


@pytest.mark.parametrize('name', ['cond', 'moeb', 'rq4', 'cfg1', 'circ'])
def test_blocked_inverse_matches_pass_per_degree_inverse(name):
    """The blocked forward-substitution inverse (row slices per degree) against the reference
    algorithm (one full conditioner pass per degree, autoregressive.py:216-227) on the same kernels."""
    g = gu.load('flows.npz')
    flow = gu.build_flow(name, g)
    yin = dev(g[f'{name}/y_f32'][:64])
    assert all(layer._blocked_ok() for layer in flow)
    xb, lb = flow.inverse(yin)
    for layer in flow:
        layer.blocked_inverse = False
    xr, lr = flow.inverse(yin)
    assert torch.allclose(xb, xr, rtol=1e-5, atol=2e-5)
    assert torch.allclose(lb, lr, rtol=1e-5, atol=1e-4)


def test_identity_initialisation():
    g = gu.load('flows.npz')
    flow = gu.build_flow('ident', g)
    x = dev(g['ident/x'])
    for fused in (True, False):
        for layer in flow:
            layer.fused = fused
        y, l = flow(x)
        assert torch.allclose(y, x, atol=1e-6)
        assert torch.allclose(l, torch.zeros_like(l), atol=1e-5)
    xi, li = flow.inverse(y)
    assert torch.allclose(xi, x, atol=1e-6) and torch.allclose(li, torch.zeros_like(li), atol=1e-5)


def test_flow_vs_oracle_medium_size():
    """Seeded D=300, 4-layer RQ-8 with the default hidden width (multi-tile GEMMs, ragged batch)
    against the CPU oracle in float64 on the same float32 weights."""
    from oracle import flows as oflows, made as omade
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D, B = 300, 777
    torch.manual_seed(0)
    layers = [MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                  transformer=NeuralSplineTransformer(torch.full((D,), -5.), torch.full((D,), 5.), 8),
                  initialize_identity=False) for i in range(2)]
    flow = SequentialFlow(*layers)
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1234)).clamp(-4.9, 4.9)
    x[5] *= 1.3                                     # a few out-of-domain values
    sd = {k: v.numpy() for k, v in flow.state_dict().items()}
    olayers = []
    for i in range(2):
        made = omade.made_layers_from_state({k: (v.astype(np.float64) if v.dtype == np.float32 else v)
                                             for k, v in sd.items()}, prefix=f'{i}._conditioner.')
        olayers.append(dict(degrees_in=omade.generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                            transformer=dict(type='spline', x0=np.full(D, -5.), xf=np.full(D, 5.), n_bins=8),
                            embedding=None, made=made))
    y_ref, l_ref = oflows.sequential_forward(x.numpy().astype(np.float64), olayers)
    flow = flow.cuda()
    for fused in (True, False):
        for layer in flow:
            layer.fused = fused
        y, l = flow(x.cuda())
        check_y(y, y_ref)
        check_ldj(l, l_ref, floor=5e-5)


# ------------------------------------------------------------------ loss / estimator

def test_loss_and_estimator():
    from tfep_amd.analysis import fep_estimator
    from tfep_amd.loss import BoltzmannKLDivLoss
    g = gu.load('loss.npz')
    uB, ldj, lw, uA = (dev(g[k]) for k in ('uB', 'ldj', 'lw', 'uA'))
    L = BoltzmannKLDivLoss()

    def close(a, key, rtol=2e-6):
        np.testing.assert_allclose(float(a), float(g[key]), rtol=rtol, atol=1e-6)
    close(L(uB, ldj), 'loss_plain_f64')
    close(L(uB, ldj, ref_potentials=uA), 'loss_ref_f64')
    close(L(uB, ldj, log_weights=lw), 'loss_weighted_f64')
    close(L(uB, ldj, log_weights=lw, ref_potentials=uA), 'loss_all_f64')
    close(L(uB), 'loss_noldj_f64')
    un = uB.clone(); un[[3, 77]] = float('nan')
    Ln = BoltzmannKLDivLoss(ignore_nan=True)
    close(Ln(un, ldj), 'loss_nan_plain_f64')
    close(Ln(un, ldj, log_weights=lw), 'loss_nan_weighted_f64')
    assert torch.isnan(L(un, ldj))                                  # NaNs propagate unless ignored (tests/test_loss.py)

    work = uB - ldj - uA
    close(fep_estimator(work), 'fep_plain_f64')
    close(fep_estimator(work * 2.5, kT=2.5), 'fep_kT_f64')
    close(fep_estimator(torch.stack([work, lw], dim=1)), 'fep_biased_f64')
    np.testing.assert_allclose(fep_estimator(dev(g['fep_vec_in_f64']), vectorized=True).cpu().numpy(),
                               g['fep_vec_f64'], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(fep_estimator(work.expand(4, -1), weights=dev(g['fep_bayes_w_f64']),
                                             vectorized=True).cpu().numpy(), g['fep_bayes_f64'], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(fep_estimator(dev(g['fep_vecb_in_f64']), vectorized=True).cpu().numpy(),
                               g['fep_vecb_f64'], rtol=2e-6, atol=1e-6)
    # statistical sanity of the reference's own test (tests/analysis/test_bootstrap.py:178-190):
    # work ~ N(0, 1)  ->  dF ~ -0.5
    w = torch.randn(200000, generator=torch.Generator().manual_seed(0)).cuda()
    assert abs(float(fep_estimator(w)) + 0.5) < 0.02


# ------------------------------------------------------------------ full-size properties (BASELINE configs 2 and 4)

def test_cfg2_size_properties():
    """One MAF + RQ-8 layer at the north-star width (D=3000, H=14998, P*D=75000): properties that do
    not need an oracle at this size -- fused == generic path, row independence (a row computed in a
    small batch equals the same row in a large one, bit for bit), blocked-inverse round trip."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D, B = 3000, 1024
    torch.manual_seed(0)
    with torch.device('cuda'):
        flow = SequentialFlow(MAF(generate_degrees(D, 'descending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    made = flow[0]._conditioner
    assert [tuple(l.mask.shape) for l in made.layers[::2]] == [(14998, 3000), (14998, 14998), (75000, 14998)]
    assert sum(int(torch.count_nonzero(l.mask)) for l in made.layers[::2]) == 697321650     # SURVEY.md 8a, descending layer
    x = torch.randn(B, D, device='cuda', generator=torch.Generator('cuda').manual_seed(1)).clamp_(-4.9, 4.9)
    y, l = flow(x)
    assert torch.isfinite(y).all() and torch.isfinite(l).all()
    # fused vs generic (parameters through HBM) path
    flow[0].fused = False
    y2, l2 = flow(x[:256])
    flow[0].fused = True
    assert float((y2 - y[:256]).detach().norm() / y[:256].detach().norm()) < 1e-6
    assert torch.allclose(l2, l[:256], rtol=1e-5, atol=1e-3)
    # split-f16 GEMMs (default) vs exact-fp32 MFMA GEMMs: fp32-equivalent, and deterministic run to run
    if os.environ.get('TFEP_SPLIT_GEMM', '1') == '0':
        return                                           # (A/B switch: what follows compares the split path with fp32)
    assert flow[0]._use_split_gemm()
    ya, la = flow(x)
    assert torch.equal(ya, y) and torch.equal(la, l)
    flow[0].split_gemm = False
    yb, lb = flow(x)
    flow[0].split_gemm = None
    assert float((yb - y).detach().norm() / yb.detach().norm()) < 1e-6
    assert float((yb - y).detach().abs().max()) < 2e-5
    assert torch.allclose(lb, l, rtol=1e-5, atol=1e-3)
    # row independence, bitwise
    y3, l3 = flow(x[100:300])
    assert torch.equal(y3, y[100:300]) and torch.equal(l3, l[100:300])
    # round trip through the blocked inverse (3000 degrees)
    xi, li = flow.inverse(y[:256])
    assert float((xi - x[:256]).norm() / x[:256].norm()) < 1e-5
    assert torch.allclose(li + l[:256], torch.zeros(256, device='cuda'), atol=2e-3)


_CFG2_ORACLE = {}          # the float64 / float32 oracle results of the layer, shared by the two arithmetic modes


@pytest.mark.parametrize('split', [True, False])
def test_cfg2_layer_vs_fp64_oracle(split):
    """One MAF + RQ-8 layer at the BASELINE cfg2 width (D=3000, H=14998, P*D=75000) on 512 rows against the float64
    oracle (same float32 weights and inputs), with the documented tolerances ASSERTED: rel L2(y) <= 1e-5 and
    per sample |d ldj| <= max(1e-5 max(1, |ldj|), 4 x the float32 oracle's own error vs float64 on the same data)
    -- for the split-f16 GEMMs (the default forward) and for the exact-fp32 MFMA GEMMs."""
    from oracle import made as omade, transformers as otr
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D, B, K = 3000, 512, 8
    torch.manual_seed(0)
    with torch.device('cuda'):
        layer = MAF(generate_degrees(D, 'ascending'),
                    transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), K),
                    initialize_identity=False)
    layer.split_gemm = split
    x = torch.randn(B, D, generator=torch.Generator().manual_seed(1234)).clamp_(-4.9, 4.9)
    with torch.no_grad():
        y, l = layer(x.cuda())
    y, l = y.cpu().numpy().astype(np.float64), l.cpu().numpy().astype(np.float64)

    def oracle(dtype):
        # oracle/made.py row-chunked over the output units (weight norm is per row): the float64 copy of the
        # 75000 x 14998 output layer would not fit otherwise
        h = x.numpy().astype(dtype)
        lins = list(layer._conditioner.layers[::2])
        for li, lin in enumerate(lins):
            outs = []
            for r0 in range(0, lin.out_features, 8192):
                sl = slice(r0, r0 + 8192)
                mask = lin.mask[sl].cpu().numpy().astype(dtype)
                w = omade.weight_norm_effective(lin.weight_g[sl].detach().cpu().numpy().astype(dtype),
                                                lin.weight_v[sl].detach().cpu().numpy().astype(dtype), mask)
                outs.append(omade.masked_linear(h, w, lin.bias[sl].detach().cpu().numpy().astype(dtype), mask))
            h = np.concatenate(outs, axis=1)
            if li + 1 < len(lins):
                h = omade.elu(h)
        return otr.spline_forward(x.numpy().astype(dtype), h, np.full(D, -5.0, dtype), np.full(D, 5.0, dtype), K)

    if 'res' not in _CFG2_ORACLE:          # (same seeds -> same layer and inputs in both parametrisations)
        _CFG2_ORACLE['res'] = (oracle(np.float64), oracle(np.float32))
    (y64, l64), (y32, l32) = _CFG2_ORACLE['res']
    noise_y = np.linalg.norm(y32 - y64) / np.linalg.norm(y64)
    noise_l = np.abs(l32.astype(np.float64) - l64).max()
    rel = np.linalg.norm(y - y64) / np.linalg.norm(y64)
    err_l = np.abs(l - l64)
    tol_l = np.maximum(1e-5 * np.maximum(1.0, np.abs(l64)), 4.0 * noise_l)
    print(f'cfg2 layer vs fp64 oracle ({"split-f16" if split else "exact-fp32"} GEMMs): rel L2(y) {rel:.2e} '
          f'(fp32 oracle {noise_y:.2e}), max |d ldj| {err_l.max():.2e} (fp32 oracle {noise_l:.2e}, |ldj| median '
          f'{np.median(np.abs(l64)):.1f}), max |dy| {np.abs(y - y64).max():.2e}')
    assert rel <= 1e-5
    assert bool((err_l <= tol_l).all()), (err_l.max(), tol_l.min())
    assert err_l.max() <= 1e-5 * max(1.0, np.abs(l64).max()) + 4.0 * noise_l


def test_cfg4_size_properties():
    """Circular RQ-8 + periodic embedding on 512 torsions, batch 131072, all 4 layers of BASELINE cfg4-i: outputs stay
    in the period, the map is periodic (x and x + period agree), and the inverse round-trips on a slice."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D, B = 512, 131072
    torch.manual_seed(0)
    with torch.device('cuda'):
        flow = SequentialFlow(*[
            MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending'),
                transformer=NeuralSplineTransformer(torch.zeros(D), torch.ones(D), 8, circular=True),
                embedding=PeriodicEmbedding(D, limits=[0.0, 1.0]), initialize_identity=False) for i in range(4)])
    x = torch.rand(B, D, device='cuda', generator=torch.Generator('cuda').manual_seed(2))
    y, l = flow(x)
    assert bool(((y >= 0) & (y <= 1)).all()) and torch.isfinite(l).all()
    yp, lp = flow(x[:4096] + 1.0)                      # one period later
    dp = (yp - y[:4096]).abs()
    dp = torch.minimum(dp, 1.0 - dp)                   # on the circle: y = 0 and y = 1 are the same point
    assert float(dp.detach().max()) < 2e-5 and torch.allclose(lp, l[:4096], atol=2e-3)
    xi, li = flow.inverse(y[:512])
    d = (xi - x[:512]).abs()
    d = torch.minimum(d, 1.0 - d)                      # distance on the circle
    assert float(d.max()) < 2e-4
    assert torch.allclose(li + l[:512], torch.zeros(512, device='cuda'), atol=5e-3)


def test_cfg4_moebius_size_properties():
    """BASELINE cfg4-ii at size: 4-layer MAF with Moebius(d=2, unit sphere) on 512 torsions as unit 2-vectors (1024
    features, ``repeats=2`` degrees), batch 131072.  Size-independent properties: outputs stay on the unit circle, row
    independence bit for bit, run-to-run determinism, and the blocked inverse round-trips with cancelling log-dets."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import MoebiusTransformer
    n_vec, B = 512, 131072
    D = 2 * n_vec
    torch.manual_seed(0)
    with torch.device('cuda'):
        flow = SequentialFlow(*[
            MAF(generate_degrees(D, 'ascending' if i % 2 == 0 else 'descending', repeats=2),
                transformer=MoebiusTransformer(dimension=2, unit_sphere=True), initialize_identity=False)
            for i in range(4)])
    theta = torch.rand(B, n_vec, device='cuda', generator=torch.Generator('cuda').manual_seed(3)) * (2 * np.pi)
    x = torch.stack([torch.cos(theta), torch.sin(theta)], dim=-1).reshape(B, D)
    with torch.no_grad():
        y, l = flow(x)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(l).all())
        r = y.reshape(B, n_vec, 2).norm(dim=-1)
        assert float((r - 1).abs().max()) < 1e-4                                  # unit circle preserved through 4 layers
        y2, l2 = flow(x)
        assert torch.equal(y, y2) and torch.equal(l, l2)
        # at this batch the 3 M-weight conditioners run on split-f16 operands (MADE.split_by_batch), small batches on the
        # exact-fp32 kernels: equal to the split format's rounding; bit for bit once the arithmetic is pinned
        assert all(layer._use_split_gemm(B) and not layer._use_split_gemm(200) for layer in flow)
        for lo, hi in ((0, 200), (77777, 77777 + 131), (B - 64, B)):
            ys, ls = flow(x[lo:hi].clone())
            assert float((ys - y[lo:hi]).abs().max()) < 2e-5 and float((ls - l[lo:hi]).abs().max()) < 2e-4, (lo, hi)
        for layer in flow:
            layer._conditioner.split_by_batch = False
        y, l = flow(x)
        for lo, hi in ((0, 200), (77777, 77777 + 131), (B - 64, B)):
            ys, ls = flow(x[lo:hi].clone())
            assert torch.equal(ys, y[lo:hi]) and torch.equal(ls, l[lo:hi]), (lo, hi)
        xi, li = flow.inverse(y[5000:5000 + 512])
        assert float((xi - x[5000:5000 + 512]).abs().max()) < 5e-4
        assert torch.allclose(li + l[5000:5000 + 512], torch.zeros(512, device='cuda'), atol=5e-3)


def test_circular_spline_fixed_points_and_periodicity():
    """tests/nn/transformers/test_spline.py:308-352: with zero shift the period boundaries map to themselves;
    outputs stay inside the period; first and last knot share a slope (C1 on the circle)."""
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    x0 = torch.tensor([0.0, -1.0, 2.0])
    xf = torch.tensor([2.0, -0.5, 5.0])
    K, B = 3, 64
    t = NeuralSplineTransformer(x0=x0, xf=xf, n_bins=K, circular=True).cuda()
    gen = torch.Generator().manual_seed(0)
    eps = 1e-6
    x = torch.cat([x0[None] + eps * (xf - x0), xf[None] - eps * (xf - x0),
                   torch.rand(B - 2, 3, generator=gen) * (xf - x0) + x0]).cuda()
    par = torch.randn(B, (3 * K + 1) * 3, generator=gen).cuda()
    y, l = t(x, par)
    assert torch.all(y > x0.cuda() - 1e-6) and torch.all(y < xf.cuda() + 1e-6)
    xi, li = t.inverse(y, par)
    d = (xi - x).abs()
    d = torch.minimum(d, (xf - x0).cuda() - d)
    assert float(d.max()) < 2e-4 and torch.allclose(l + li, torch.zeros_like(l), atol=1e-4)   # fp32 y through 1/slope
    par0 = par.reshape(B, 3 * K + 1, 3).clone()
    par0[:, 3 * K] = 0.0                                         # zero shift
    y0, _ = t(x, par0.reshape(B, -1))
    assert torch.allclose(y0[:2], x[:2], atol=2e-5)              # boundaries are fixed points
    # derivative continuity across the period boundary: dy/dx just inside both ends agree
    # (log-derivative of a single feature: feed one-feature transformers)
    t1 = NeuralSplineTransformer(x0=x0[:1], xf=xf[:1], n_bins=K, circular=True).cuda()
    p1 = par0[:1, :, :1].reshape(1, -1)
    _, la = t1(x0[:1][None].cuda() + 1e-6, p1)
    _, lb = t1(xf[:1][None].cuda() - 1e-6, p1)
    assert abs(float(la - lb)) < 1e-4


@pytest.mark.parametrize('name', ['cfg1', 'rq4', 'moeb', 'circ'])
def test_fused_inverse_block_kernel_matches_per_step_launches(name):
    """tfep_inverse_block (one kernel per block of degrees, split-K block GEMMs) against the same blocked algorithm
    launched step by step; ragged batch (dead lanes in the last wave)."""
    g = gu.load('flows.npz')
    flow = gu.build_flow(name, g)
    yin = dev(g[f'{name}/y_f32'][:77])
    with torch.no_grad():
        for layer in flow:
            assert layer._blocked_ok()
            assert layer._blocked_plan(yin.device)['fused'] is not None
        xf, lf = flow.inverse(yin)
        for layer in flow:
            layer.fused_inverse = False
            layer._dev.clear()
        xs, ls = flow.inverse(yin)
        for layer in flow:
            assert layer._blocked_plan(yin.device)['fused'] is None
    assert torch.allclose(xf, xs, rtol=1e-5, atol=2e-5)
    assert torch.allclose(lf, ls, rtol=1e-5, atol=1e-4)
    # deterministic
    with torch.no_grad():
        for layer in flow:
            layer.fused_inverse = True
            layer._dev.clear()
        xf2, lf2 = flow.inverse(yin)
    assert torch.equal(xf, xf2) and torch.equal(lf, lf2)


def test_cfg2_full_batch_properties():
    """BASELINE cfg2 at its full batch (65 536 x 3000, one of the four layers): size-independent properties -- every row
    of the big batch equals, bit for bit, the same row pushed through in a small batch; run-to-run determinism; the
    blocked inverse undoes the forward on a slice; finite outputs."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D, B = 3000, 65536
    torch.manual_seed(0)
    with torch.device('cuda'):
        flow = SequentialFlow(MAF(generate_degrees(D, 'ascending'),
                                  transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                  initialize_identity=False))
    x = torch.randn(B, D, device='cuda', generator=torch.Generator('cuda').manual_seed(7)).clamp_(-4.9, 4.9)
    with torch.no_grad():
        y, l = flow(x)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(l).all())
        y2, l2 = flow(x)
        assert torch.equal(y, y2) and torch.equal(l, l2)
        for lo, hi in ((0, 256), (12345, 12345 + 77), (B - 300, B)):
            ys, ls = flow(x[lo:hi].clone())
            assert torch.equal(ys, y[lo:hi]) and torch.equal(ls, l[lo:hi]), (lo, hi)
        xi, li = flow.inverse(y[40000:40000 + 192])
        assert float((xi - x[40000:40000 + 192]).norm() / x[40000:40000 + 192].norm()) < 1e-5
        assert torch.allclose(li + l[40000:40000 + 192], torch.zeros(192, device='cuda'), atol=2e-3)


def test_cfg2_four_layer_composition():
    """BASELINE config 2 as ``bench.py`` builds it (4 MAF + RQ-8 layers, alternating degree order, D = 3000, default hidden
    width) -- the whole composition under pytest, not only one layer at a time: ``SequentialFlow`` equals its layers applied
    one after the other with the log-dets summed (sequential.py:50-68), bit for bit; rows are independent of the batch they
    sit in; the default (split) arithmetic agrees with the exact-fp32 kernels through all four layers; a 64-row slice goes
    back through the four blocked inverses (12 000 sequential degrees) to where it started; the TFEP estimator of the mapped
    work is finite."""
    import importlib.util
    import os
    from tfep_amd.analysis import fep_estimator
    spec = importlib.util.spec_from_file_location('bench', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    D, B = 3000, 2048
    flow = bench.build_flow(D, 4, 8, torch.device('cuda'))
    x = torch.randn(B, D, device='cuda', generator=torch.Generator('cuda').manual_seed(5)).clamp_(-4.9, 4.9)
    with torch.no_grad():
        y, ldj = flow(x)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(ldj).all())
        z, tot = x, torch.zeros(B, device='cuda')
        for layer in flow:
            z, l = layer(z)
            tot = tot + l
        assert torch.equal(z, y) and torch.allclose(tot, ldj, rtol=0, atol=1e-4 * float(ldj.abs().max()))
        y2, l2 = flow(x[300:900])
        assert torch.equal(y2, y[300:900]) and torch.equal(l2, ldj[300:900])
        if all(layer._use_split_gemm(B) for layer in flow):
            for layer in flow:
                layer.split_gemm = False
            ye, le = flow(x[:512])
            for layer in flow:
                layer.split_gemm = None
            assert float((ye - y[:512]).norm() / ye.norm()) < 2e-6
            assert float((le - ldj[:512]).abs().max()) < 1e-5 * max(1.0, float(le.abs().max())) + 2e-4
        xb, lb = flow.inverse(y[:64])
        assert float((xb - x[:64]).norm() / x[:64].norm()) < 2e-5
        assert float((lb + ldj[:64]).abs().max()) < 5e-3
        dF = fep_estimator(torch.randn(B, device='cuda') - ldj)
        assert bool(torch.isfinite(dF))
