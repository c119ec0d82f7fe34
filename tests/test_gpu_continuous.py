"""Config 5 on the GPU: EGNN dynamics (velocity, Jacobian-vector products), trace estimators and the continuous flow
through the C ABI, against goldens generated from the reference (``tests/golden/continuous.npz``) and the float64
oracle (``oracle/egnn.py``)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from oracle import egnn as oe

pytestmark = pytest.mark.gpu

CONFIGS = ['tiny', 'cutoff', 'default', 'pair']
REL = 1e-5


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().cuda()


def build_dynamics(name, g, split=None):
    from tfep_amd.nn.dynamics import EGNNDynamics
    cfg = gu.continuous_configs()[name]
    kw = {k: cfg[k] for k in ('node_types', 'r_cutoff', 'time_feat_dim', 'node_feat_dim', 'distance_feat_dim', 'n_layers',
                              'speed_factor')}
    dyn = EGNNDynamics(initialize_identity=False, **kw)
    sd = {k: v.float() if v.is_floating_point() else v for k, v in gu.continuous_state(g, name, torch.float32).items()}
    dyn.load_state_dict(sd, strict=True)
    dyn.split_gemm = split              # None: the default (split-f16 edge products); False: exact-fp32 MFMA
    return dyn.cuda(), cfg


def rel_l2(got, ref):
    got = got.detach().cpu().numpy().astype(np.float64)
    return float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300))


@pytest.mark.parametrize('split', [True, False])
@pytest.mark.parametrize('name', CONFIGS)
def test_dynamics_velocity_matches_reference(name, split):
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics(name, g, split)
    x, t = dev(g[f'{name}/x']), float(g[f'{name}/t'][0])
    with torch.no_grad():
        vel = dyn(torch.tensor(t), x)
    ref = g[f'{name}/vel_f64']
    noise = np.linalg.norm(g[f'{name}/vel_f32'] - ref) / np.linalg.norm(ref)
    r = rel_l2(vel, ref)
    print(f'{name} ({"split-f16" if split else "exact-fp32"} edge products): velocity rel L2 {r:.2e} '
          f'(reference float32 vs float64: {noise:.2e})')
    assert r <= REL
    # deterministic, and independent of the batch a sample sits in
    with torch.no_grad():
        assert torch.equal(dyn(torch.tensor(t), x), vel)
        assert torch.equal(dyn(torch.tensor(t), x[1:3].clone()), vel[1:3])


@pytest.mark.parametrize('split', [True, False])
@pytest.mark.parametrize('name', CONFIGS)
def test_jvp_and_trace_match_reference_jacobian(name, split):
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics(name, g, split)
    x, t, eps = dev(g[f'{name}/x']), float(g[f'{name}/t'][0]), dev(g[f'{name}/eps'])
    jac = g[f'{name}/jacobian_f64']
    B = x.shape[0]
    trace = torch.zeros(B, device='cuda')
    frob = torch.zeros(B, device='cuda')
    vsq = torch.zeros(B, device='cuda')
    with torch.no_grad():
        vel, jv = dyn.jvp(t, x, eps[0], trace=trace, frobenius=frob, scale=1.0, velocity_squared_norm=vsq)
        vel0 = dyn(t, x)
    assert torch.equal(vel, vel0)                                        # the tangent does not perturb the primal
    e64 = g[f'{name}/eps'][0].astype(np.float64)
    jv_ref = np.einsum('bij,bj->bi', jac, e64)
    assert rel_l2(jv, jv_ref) <= REL
    tr_ref = g[f'{name}/hutchinson1_reg/trace_f64']                      # (e^T J) . e of the reference == e . (J e)
    scale = np.abs(jv_ref * e64).sum(-1)                                 # the terms the trace sums
    assert np.all(np.abs(trace.cpu().numpy() - tr_ref) <= 1e-5 * np.maximum(scale, 1.0))
    np.testing.assert_allclose(frob.cpu().numpy(), (jv_ref ** 2).sum(-1), rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(vsq.cpu().numpy(), (g[f'{name}/vel_f64'] ** 2).sum(-1), rtol=1e-4, atol=1e-7)
    # exact trace = sum_k e_k . (J e_k): every column of the Jacobian through the kernel
    D = x.shape[1]
    tr = torch.zeros(B, device='cuda')
    fr = torch.zeros(B, device='cuda')
    cols = []
    with torch.no_grad():
        for k in range(D):
            e = torch.zeros(B, D, device='cuda')
            e[:, k] = 1.0
            cols.append(dyn.jvp(t, x, e, trace=tr, frobenius=fr)[1])
    jac_gpu = torch.stack(cols, dim=2)                                   # [b, i, k] = d vel_i / d x_k
    assert rel_l2(jac_gpu, jac) <= REL
    ex = g[f'{name}/exact1_reg/trace_f64']
    assert np.all(np.abs(tr.cpu().numpy() - ex) <= 1e-5 * np.maximum(np.abs(np.diagonal(jac, axis1=1, axis2=2)).sum(-1), 1.0))
    np.testing.assert_allclose(fr.cpu().numpy(), (jac ** 2).sum((1, 2)), rtol=1e-4, atol=1e-7)


def test_identity_initialisation_and_state_dict_schema():
    from tfep_amd.nn.dynamics import EGNNDynamics
    g = gu.load('continuous.npz')
    torch.manual_seed(0)
    dyn = EGNNDynamics(node_types=[0, 1, 0], r_cutoff=5.0, time_feat_dim=2, node_feat_dim=4, distance_feat_dim=3, n_layers=2)
    ref_sd = {k[len('identity/sd/'):]: g[k] for k in g.files if k.startswith('identity/sd/')}
    sd = dyn.state_dict()
    assert list(sd.keys()) == list(ref_sd.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == ref_sd[k].shape, k
        # same construction order + same seed = the reference's initial parameters
        np.testing.assert_allclose(v.numpy(), ref_sd[k], rtol=0, atol=0, err_msg=k)
    dyn = dyn.cuda()
    x = torch.randn(4, 9, generator=torch.Generator().manual_seed(5)).cuda()
    with torch.no_grad():
        assert float(dyn(0.3, x).abs().max()) == 0.0                     # zero velocity (egnn.py:136-138)


def test_equivariance_properties():
    """tests/nn/dynamics/test_egnn.py:83-160: rotations rotate the velocity, translations leave it unchanged,
    permuting two nodes of the same type permutes their velocities."""
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics('cutoff', g)
    x = dev(g['cutoff/x'])
    B, n = x.shape[0], len(cfg['node_types'])
    t = 0.37
    q, _ = torch.linalg.qr(torch.randn(3, 3, generator=torch.Generator().manual_seed(3), dtype=torch.float64))
    if torch.det(q) < 0:
        q[:, 0] = -q[:, 0]
    R = q.float().cuda()
    with torch.no_grad():
        vel = dyn(t, x)
        vel_rot = dyn(t, (x.reshape(B, n, 3) @ R.T).reshape(B, -1))
        assert torch.allclose(vel_rot, (vel.reshape(B, n, 3) @ R.T).reshape(B, -1), atol=2e-5)
        shift = torch.randn(B, 1, 3, device='cuda')
        assert torch.allclose(dyn(t, (x.reshape(B, n, 3) + shift).reshape(B, -1)), vel, atol=2e-5)
        i, j = 1, 2                                                      # both of type 1 in the 'cutoff' fixture
        perm = list(range(n)); perm[i], perm[j] = j, i
        xp = x.reshape(B, n, 3)[:, perm].reshape(B, -1)
        assert torch.allclose(dyn(t, xp), vel.reshape(B, n, 3)[:, perm].reshape(B, -1), atol=2e-5)
        # centre of geometry is preserved: the velocities of a sample sum to zero
        assert float(vel.reshape(B, n, 3).sum(1).abs().max()) < 1e-5


def test_radial_and_segment_sum_helpers():
    from tfep_amd.nn import graph
    from tfep_amd.nn.embeddings import BehlerParrinelloRadialExpansion, GaussianBasisExpansion
    g = gu.load('continuous.npz')
    r = dev(g['radial/r'])
    gb = GaussianBasisExpansion.from_range(n_gaussians=5, max_mean=1.0, trainable_stds=True).cuda()
    bp = BehlerParrinelloRadialExpansion.from_range(r_cutoff=2.0, n_gaussians=6, max_mean=2.0, trainable_stds=True).cuda()
    bp2 = BehlerParrinelloRadialExpansion(r_cutoff=2.0, means=torch.tensor([0.1, 0.7, 1.9]), stds=torch.tensor([0.3, 0.2, 0.5]),
                                          force_zero_after_cutoff=False).cuda()
    np.testing.assert_allclose(gb(r).cpu().numpy(), g['radial/gauss_f64'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(bp(r).cpu().numpy(), g['radial/bp_f64'], rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(bp2(r).cpu().numpy(), g['radial/bp_nozero_f64'], rtol=2e-6, atol=2e-7)
    assert gb(r.reshape(5, 9)).shape == (5, 9, 5) and gb(r.reshape(5, 9, 1)).shape == (5, 9, 5)
    assert list(gb.state_dict().keys()) == ['_log_gammas'] and list(bp2.state_dict().keys()) == []
    e = graph.get_all_edges(2, 4)
    out = graph.unsorted_segment_sum(dev(g['graph/seg_data']), e[1], 8)
    np.testing.assert_allclose(out.cpu().numpy(), g['graph/seg_sum'], rtol=1e-6, atol=1e-6)
    assert graph.unsorted_segment_sum(torch.zeros(0, 3, device='cuda'), torch.zeros(0, dtype=torch.int64), 4).abs().sum() == 0


# ------------------------------------------------------------------ the continuous flow

def oracle_dynamics(name, g):
    cfg = gu.continuous_configs()[name]
    sd = gu.continuous_state(g, name)
    return (lambda t, x: oe.egnn_dynamics(t, x, sd, cfg)), cfg


@pytest.mark.parametrize('name,method,n_steps', [('tiny', 'rk4', 5), ('cutoff', 'rk4', 4), ('cutoff', 'midpoint', 8),
                                                 ('pair', 'euler', 10), ('default', 'rk4', 2)])
@pytest.mark.parametrize('estimator', ['hutchinson', 'exact'])
def test_flow_matches_float64_oracle_on_the_same_grid(name, method, n_steps, estimator):
    """ContinuousFlow over the HIP dynamics against the float64 oracle integrating with the same scheme, grid and
    Hutchinson noise (the ODE stepping has no reference golden -- torchdiffeq is absent: parity unpinned -- but the
    integrands are pinned by the goldens, see test_jvp_and_trace_match_reference_jacobian)."""
    from tfep_amd.nn.flows import ContinuousFlow
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics(name, g)
    odyn, _ = oracle_dynamics(name, g)
    x32 = g[f'{name}/x'][:4]
    eps32 = g[f'{name}/eps'][:2, :4]
    if estimator == 'exact' and name == 'default':
        pytest.skip('36 coordinates x 4 layers x 8 stages in the float64 oracle: covered by the smaller fixtures')
    flow = ContinuousFlow(dyn, trace_estimator=estimator, solver=method, solver_options={'step_size': 1.0 / n_steps},
                          n_hutchinson_samples=2, regularization=True)
    flow.ode_func.fixed_noise = dev(eps32)
    with torch.no_grad():
        y, tr, reg = flow(dev(x32))
    assert flow.last_solver_stats['n_steps'] == n_steps
    # (the oracle in the reference's reverse-mode form: with the regulariser the flow runs the reverse-pass kernels)
    yo, tro, rego = oe.continuous_flow(odyn, torch.from_numpy(x32).double(), n_steps, method, estimator=estimator,
                                       eps=torch.from_numpy(eps32).double(), regularization=True, frobenius_from='vjp')
    assert rel_l2(y, yo.numpy()) <= REL
    np.testing.assert_allclose(tr.cpu().numpy(), tro.numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(reg.cpu().numpy(), rego.numpy(), rtol=1e-4, atol=2e-5)
    # without the regularisation term the state has two components, the trace comes from the forward-mode kernel
    # (e . (J e) instead of (e^T J) . e: the same number up to rounding) and y is unchanged
    flow.regularization = False
    with torch.no_grad():
        y2, tr2 = flow(dev(x32))
    assert torch.equal(y2, y) and torch.allclose(tr2, tr, rtol=1e-4, atol=2e-5)
    if estimator == 'hutchinson':
        flow.ode_func.reverse_mode = False                # forward mode with the regulariser: |J e|^2 instead of |e^T J|^2
        flow.regularization = True
        with torch.no_grad():
            y3, tr3, reg3 = flow(dev(x32))
        _, _, rego_f = oe.continuous_flow(odyn, torch.from_numpy(x32).double(), n_steps, method, estimator=estimator,
                                          eps=torch.from_numpy(eps32).double(), regularization=True, frobenius_from='jvp')
        assert torch.equal(y3, y) and torch.allclose(tr3, tr, rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(reg3.cpu().numpy(), rego_f.numpy(), rtol=1e-4, atol=2e-5)
        flow.ode_func.reverse_mode = None
        flow.regularization = False
    # inverse: integrates back from t = 1, the trace comes out negated (continuous.py:176-180)
    with torch.no_grad():
        xb, trb = flow.inverse(y)
    xo, tbo = oe.continuous_flow(odyn, yo, n_steps, method, inverse=True, estimator=estimator,
                                 eps=torch.from_numpy(eps32).double(), regularization=False)
    assert rel_l2(xb, xo.numpy()) <= REL
    np.testing.assert_allclose(trb.cpu().numpy(), tbo.numpy(), rtol=1e-4, atol=2e-5)


def test_flow_round_trip_and_adaptive_solver():
    """tests/nn/flows/test_continuous.py:221-260 on the HIP dynamics: inverse(forward(x)) == x with cancelling traces
    (exact estimator), for the fixed-grid rk4 and the adaptive dopri5; dopri5 agrees with a fine rk4 grid to its
    tolerance and reports its step counts."""
    from tfep_amd.nn.flows import ContinuousFlow
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics('cutoff', g)
    x = dev(g['cutoff/x'])
    fine = ContinuousFlow(dyn, trace_estimator='exact', solver='rk4', solver_options={'step_size': 1 / 32}, regularization=False)
    with torch.no_grad():
        y_ref, tr_ref = fine(x)
        xb, trb = fine.inverse(y_ref)
    assert float((xb - x).abs().max()) < 2e-5 and float((tr_ref + trb).abs().max()) < 2e-4
    flow = ContinuousFlow(dyn, trace_estimator='exact', regularization=False)          # solver='dopri5', the default
    with torch.no_grad():
        y, tr = flow(x)
        stats = dict(flow.last_solver_stats)
        xi, tri = flow.inverse(y)
    assert stats['n_steps'] >= 1 and stats['n_evaluations'] >= 7
    # rtol = atol = 1e-4 on an RMS norm over the whole batch: individual coordinates end a few 1e-3 off (the reference's
    # own round-trip test uses atol 1e-3 on a much smoother toy dynamics)
    # (with the error weights of torchdiffeq's tableau -- Shampine's b* -- the accepted steps are a little longer than with
    # Dormand & Prince's own: 6.7e-3 where those gave < 5e-3)
    assert float((y - y_ref).abs().max()) < 1e-2 and float((tr - tr_ref).abs().max()) < 2e-2
    assert float((xi - x).abs().max()) < 1e-2 and float((tr + tri).abs().max()) < 2e-2
    # fresh noise per integration unless fixed (continuous.py:223-229)
    hut = ContinuousFlow(dyn, solver='euler', solver_options={'step_size': 0.5}, regularization=False)
    with torch.no_grad():
        _, t1 = hut(x)
        _, t2 = hut(x)
        hut.ode_func.fixed_noise = torch.randn(1, *x.shape, device='cuda')
        _, t3 = hut(x)
        _, t4 = hut(x)
    assert not torch.equal(t1, t2) and torch.equal(t3, t4)


def test_flow_api_schema_and_errors():
    from tfep_amd.nn.dynamics import EGNNDynamics
    from tfep_amd.nn.flows import ContinuousFlow
    dyn = EGNNDynamics([0, 1], r_cutoff=3.0, time_feat_dim=2, node_feat_dim=4, distance_feat_dim=3, n_layers=1)
    flow = ContinuousFlow(dyn)
    assert all(k.startswith('ode_func.dynamics.') for k in flow.state_dict())
    assert flow.ode_func.trace_estimator == 'hutchinson' and flow.solver == 'dopri5' and flow.regularization
    with pytest.raises(ValueError):
        ContinuousFlow(dyn, trace_estimator='foo')
    flow = flow.cuda()
    x = torch.randn(3, 6, device='cuda')
    y, tr, reg = flow(x)                                      # identity initialisation: nothing moves
    assert torch.equal(y.detach(), x) and float(tr.abs().max()) == 0.0 and float(reg.abs().max()) == 0.0
    (y.sum() + tr.sum() + reg.sum()).backward()              # trainable: under grad mode the flow takes the autograd route
    grads = [p.grad for p in flow.parameters() if p.grad is not None]
    assert grads and all(torch.isfinite(gr).all() for gr in grads)
    with pytest.raises(ValueError):
        ContinuousFlow(dyn, solver='adams')(x.detach())
    with pytest.raises(_lib_error()):
        flow(x.cpu())                                        # no CPU fallback
    # the kernels stop at 64 features; wider dynamics run on the torch route (with or without grad mode), a flow over
    # them on autograd like any torch dynamics; the kernel-only products say so loudly
    wide = EGNNDynamics([0, 1], r_cutoff=3.0, node_feat_dim=65, initialize_identity=False).cuda()
    assert not wide.kernels_supported() and dyn.cuda().kernels_supported()
    with torch.no_grad():
        v0 = wide(0.0, x.detach())
        yw, tw = ContinuousFlow(wide, solver='rk4', solver_options={'step_size': 0.5}, regularization=False)(x.detach())
    assert v0.shape == x.shape and torch.allclose(wide(0.0, x.detach()).detach(), v0, rtol=1e-5, atol=1e-6)
    assert yw.shape == x.shape and bool(torch.isfinite(tw).all())
    with pytest.raises(NotImplementedError):
        wide.jvp(0.0, x.detach(), torch.ones_like(x))


def _lib_error():
    from tfep_amd._lib import TfepHipError
    return TfepHipError


class DynamicsMLP(torch.nn.Module):
    """A user-supplied torch dynamics (the one the reference's own tests use, test_continuous.py:82-103)."""

    def __init__(self, n_features):
        super().__init__()
        self.mlp = torch.nn.Sequential(
            torch.nn.Linear(n_features + 1, n_features), torch.nn.SiLU(), torch.nn.Linear(n_features, n_features),
            torch.nn.SiLU(), torch.nn.Linear(n_features, n_features, bias=False), torch.nn.Tanh())

    def forward(self, t, x):
        if len(x.shape) > 1:
            t = t.reshape(1, 1).expand(x.shape[0], 1)
        return self.mlp(torch.cat([t, x], dim=-1))


def test_user_supplied_torch_dynamics_through_autograd():
    """Any ``dynamics(t, x)`` module still works: velocity and vector-Jacobian products by autograd on
    the device.  Exact trace / Frobenius norm against the full autograd Jacobian, Hutchinson in expectation, round trip,
    identity flow with a stable backward (test_continuous.py:125-260)."""
    from tfep_amd.nn.flows import ContinuousFlow
    torch.manual_seed(2)
    D, B = 9, 5
    dyn = DynamicsMLP(D).cuda().double()
    x = torch.randn(B, D, device='cuda', dtype=torch.float64)
    t = torch.tensor(0.3, device='cuda', dtype=torch.float64)
    jac = torch.stack([torch.autograd.functional.jacobian(lambda z: dyn(t, z[None])[0], x[b]) for b in range(B)])
    flow = ContinuousFlow(dyn, trace_estimator='exact', solver='rk4', solver_options={'step_size': 0.1}, requires_backward=False)
    vel, tr, reg = flow.ode_func(t, (x, x.new_zeros(B), x.new_zeros(B)))
    assert torch.allclose(tr, torch.diagonal(jac, dim1=1, dim2=2).sum(-1))
    assert torch.allclose(reg, (vel ** 2).sum(-1) + (jac ** 2).sum((1, 2)))
    hut = ContinuousFlow(dyn, n_hutchinson_samples=4000, requires_backward=False)
    hut.ode_func.before_odeint(x)
    _, trh, regh = hut.ode_func(t, (x, x.new_zeros(B), x.new_zeros(B)))
    assert torch.allclose(trh, tr, atol=5e-2) and torch.allclose(regh, reg, atol=5e-2)
    with torch.no_grad():
        y, tr1, _ = flow(x)
        xb, tr2, _ = flow.inverse(y)
    assert torch.allclose(xb, x, atol=1e-5) and torch.allclose(tr1 + tr2, torch.zeros_like(tr1), atol=1e-5)
    # identity dynamics: y == x, zero trace, gradient of sum(y) w.r.t. x is one (test_identity_flow)
    dyn.mlp[-2].weight.data.fill_(0.0)
    xg = x.clone().requires_grad_(True)
    y, tr, reg = ContinuousFlow(dyn, trace_estimator='exact', solver='rk4', solver_options={'step_size': 0.25})(xg)
    assert torch.allclose(y, xg) and torch.allclose(tr, torch.zeros_like(tr))
    y.sum().backward()
    assert torch.allclose(xg.grad, torch.ones_like(xg.grad))


def test_cfg5_full_size_properties():
    """BASELINE config 5 at size: 3 x 256 atoms, batch 16384, 10 rk4 steps, Hutchinson trace, default EGNN widths (4
    layers, 64 features, 64 radial functions), every pair inside the cutoff.  Size-independent properties: finite outputs;
    the centre of geometry of every sample is preserved (egnn.py:187-191); a slice pushed through alone (same noise rows)
    reproduces its rows of the big batch bit for bit (rows are independent, the reductions have a fixed order); the
    inverse integrates a slice back to its start with cancelling traces; the split-f16 edge products agree with the
    exact-fp32 MFMA chain on a slice."""
    from tfep_amd.nn.dynamics import EGNNDynamics
    from tfep_amd.nn.flows import ContinuousFlow
    B, n = 16384, 256
    gen = torch.Generator(device='cuda').manual_seed(7)
    side, a = 7, 0.215                                                   # jittered lattice at ~100 atoms / nm^3
    grid = torch.stack(torch.meshgrid(*[torch.arange(side, dtype=torch.float32)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n]
    x = (grid.cuda()[None] * a + (torch.rand(B, n, 3, device='cuda', generator=gen) - 0.5) * 0.3 * a).reshape(B, 3 * n)
    torch.manual_seed(0)
    # (randomly initialised weights and 255 neighbours per atom: speed_factor keeps the velocity field smooth enough for
    # 10 rk4 steps to resolve, so that the round trip below tests the kernels and not the truncation error)
    dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=4.0, speed_factor=0.02,
                       initialize_identity=False).cuda()
    flow = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 0.1}, regularization=False)
    eps = torch.randn(1, B, 3 * n, device='cuda', generator=gen)
    flow.ode_func.fixed_noise = eps
    with torch.no_grad():
        y, tr = flow(x)
        assert flow.last_solver_stats == dict(n_steps=10, n_rejected=0, n_evaluations=40)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(tr).all())
        assert float((y - x).abs().max()) > 1e-4                         # the flow moves the atoms ...
        c0, c1 = x.reshape(B, n, 3).mean(1), y.reshape(B, n, 3).mean(1)
        assert float((c1 - c0).abs().max()) < 2e-5                       # ... but not a sample's centre of geometry
        for lo, hi in ((0, 48), (9000, 9000 + 33)):
            flow.ode_func.fixed_noise = eps[:, lo:hi].clone()
            ys, ts = flow(x[lo:hi].clone())
            assert torch.equal(ys, y[lo:hi]) and torch.equal(ts, tr[lo:hi]), (lo, hi)
        lo, hi = 4000, 4000 + 128
        flow.ode_func.fixed_noise = eps[:, lo:hi].clone()
        xb, tb = flow.inverse(y[lo:hi].clone())
        assert float((xb - x[lo:hi]).abs().max()) < 1e-4                 # rk4 forward then backward on the same grid
        assert float((tb + tr[lo:hi]).abs().max()) < 1e-2 * max(1.0, float(tr[lo:hi].abs().max()))
        dyn.split_gemm = False
        ye, te = flow(x[lo:hi].clone())
        dyn.split_gemm = None
        assert float((ye - y[lo:hi]).norm() / y[lo:hi].norm()) < 1e-6
        assert float((te - tr[lo:hi]).abs().max()) < 1e-3 * max(1.0, float(tr[lo:hi].abs().max()))


@pytest.mark.parametrize('split', [True, False])
@pytest.mark.parametrize('name', CONFIGS)
def test_vjp_matches_reference_reverse_mode(name, split):
    """``EGNNDynamics.vjp``: e^T J by the reverse-pass kernels against the reference's full Jacobian, and the quadratic
    forms the reference's Hutchinson estimators build from it -- trace (e^T J) . e and |e^T J|^2 -- against its own
    ``_trace_and_frobenious_squared_norm_hutchinson`` goldens (continuous.py:344-361), 1 and 3 noise samples."""
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics(name, g, split)
    x, t, eps = dev(g[f'{name}/x']), float(g[f'{name}/t'][0]), dev(g[f'{name}/eps'])
    jac = g[f'{name}/jacobian_f64']
    B = x.shape[0]
    e64 = g[f'{name}/eps'].astype(np.float64)
    with torch.no_grad():
        vel0 = dyn(t, x)
        for n_s, key in ((1, 'hutchinson1_reg'), (3, 'hutchinson3_reg')):
            trace, frob, vsq = (torch.zeros(B, device='cuda') for _ in range(3))
            for s_ in range(n_s):
                vel, ej = dyn.vjp(t, x, eps[s_], trace=trace, frobenius=frob, scale=1.0 / n_s,
                                  velocity_squared_norm=vsq if s_ == 0 else None)
                assert torch.equal(vel, vel0)
                ref = np.einsum('bi,bij->bj', e64[s_], jac)
                assert rel_l2(ej, ref) <= REL, (name, s_, rel_l2(ej, ref))
            scale = np.mean([np.abs(np.einsum('bi,bij->bj', e64[s_], jac) * e64[s_]).sum(-1) for s_ in range(n_s)], axis=0)
            assert np.all(np.abs(trace.cpu().numpy() - g[f'{name}/{key}/trace_f64']) <= 1e-5 * np.maximum(scale, 1.0))
            reg = (vsq + frob).cpu().numpy()
            np.testing.assert_allclose(reg, g[f'{name}/{key}/reg_f64'], rtol=1e-4, atol=1e-6)
        # the forward-mode and reverse-mode traces are the same number
        tr_f = torch.zeros(B, device='cuda')
        tr_r = torch.zeros(B, device='cuda')
        dyn.jvp(t, x, eps[0], trace=tr_f, need_jvp=False)
        ej1 = dyn.vjp(t, x, eps[0], trace=tr_r)[1]
        assert torch.allclose(tr_f, tr_r, rtol=1e-4, atol=1e-4)
        # deterministic, and independent of the batch a sample sits in
        assert torch.equal(dyn.vjp(t, x, eps[0])[1], ej1)
        if B > 2:
            assert torch.equal(dyn.vjp(t, x[1:3].clone(), eps[0, 1:3].clone())[1], ej1[1:3])


@pytest.mark.parametrize('solver,options,tol', [('bosh3', None, 5e-3), ('adaptive_heun', None, 5e-3), ('fehlberg2', None, 5e-2),
                                                ('heun3', {'step_size': 1 / 16}, 1e-4), ('midpoint', {'step_size': 1 / 64}, 1e-4)])
def test_other_solvers_agree_with_a_fine_rk4_grid(solver, options, tol):
    from tfep_amd.nn.flows import ContinuousFlow
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics('tiny', g)
    x = dev(g['tiny/x'])
    eps = dev(g['tiny/eps'][:1])
    ref = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 1 / 64}, regularization=False)
    flow = ContinuousFlow(dyn, solver=solver, solver_options=options, regularization=False)
    ref.ode_func.fixed_noise = flow.ode_func.fixed_noise = eps
    with torch.no_grad():
        y0, t0 = ref(x)
        y1, t1 = flow(x)
    assert float((y1 - y0).abs().max()) < tol and float((t1 - t0).abs().max()) < 10 * tol
    assert flow.last_solver_stats['n_steps'] >= 1


def test_empty_batch_through_the_dynamics_and_the_flow():
    """B = 0: empty velocity / tangent / cotangent and an empty flow output, no launch on an empty grid."""
    from tfep_amd.nn.dynamics import EGNNDynamics
    from tfep_amd.nn.flows import ContinuousFlow
    dyn = EGNNDynamics(node_types=[0, 1, 1, 0], r_cutoff=2.0, time_feat_dim=4, node_feat_dim=16, distance_feat_dim=8, n_layers=2,
                       initialize_identity=False).cuda()
    x = torch.empty(0, 12, device='cuda')
    with torch.no_grad():
        assert dyn(0.3, x).shape == (0, 12)
        vel, jv = dyn.jvp(0.3, x, x.clone(), trace=torch.empty(0, device='cuda'))
        assert vel.shape == (0, 12) and jv.shape == (0, 12)
        vel, gj = dyn.vjp(0.3, x, x.clone())
        assert vel.shape == (0, 12) and gj.shape == (0, 12)
        for reg in (False, True):
            flow = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 0.5}, regularization=reg)
            out = flow(x)
            assert len(out) == (3 if reg else 2) and out[0].shape == (0, 12) and all(o.shape == (0,) for o in out[1:])
            assert flow.inverse(out[0])[0].shape == (0, 12)


# ------------------------------------------------------------------ training: the differentiable route

@pytest.mark.parametrize('name', ['tiny', 'cutoff', 'pair', 'default'])
def test_torch_route_equals_the_kernels_and_the_reference(name):
    """``EGNNDynamics.torch_forward`` (what runs under grad mode) against the kernels and the reference's float64 velocity."""
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics(name, g, split=False)
    x, t = dev(g[f'{name}/x']), float(g[f'{name}/t'][0])
    with torch.no_grad():
        vel_k = dyn(torch.tensor(t), x)
    xg = x.clone().requires_grad_(True)
    vel_t = dyn(torch.tensor(t), xg)
    assert vel_t.requires_grad
    assert rel_l2(vel_t, g[f'{name}/vel_f64']) <= 2e-6
    assert rel_l2(vel_t, vel_k.cpu().numpy().astype(np.float64)) <= 2e-6
    # and in float64 (the route is plain torch): the golden to round-off
    dyn64 = dyn.double()
    vel64 = dyn64(torch.tensor(t, dtype=torch.float64), x.double().requires_grad_(True))
    # (the fixed Gaussian means are float32-rounded buffers here; the reference built them in float64: 1e-8 relative)
    assert rel_l2(vel64, g[f'{name}/vel_f64']) <= 2e-7


@pytest.mark.parametrize('estimator,n_hut', [('hutchinson', 2), ('exact', 1)])
@pytest.mark.parametrize('name', ['tiny', 'cutoff', 'pair'])
def test_integrand_gradients_match_reference_double_backward(name, estimator, n_hut):
    """What a training step differentiates at every solver stage: loss = <a, vel> + <b, trace> + <c, reg> of the ODE
    function's integrands (continuous.py:231-278 with ``create_graph``), gradients of x and of every parameter against the
    reference's float64 autograd (tests/golden/continuous_grads.npz) -- second derivatives of the dynamics.  Float32 on the
    device and float64 on the device."""
    from tfep_amd.nn.flows import ContinuousFlow
    g, gg = gu.load('continuous.npz'), gu.load('continuous_grads.npz')
    for dt, tol in ((torch.float32, 5e-4), (torch.float64, 1e-6)):        # (float64: float32-rounded fixed buffers, see above)
        dyn, cfg = build_dynamics(name, g, split=False)
        dyn = dyn.to(dt)
        flow = ContinuousFlow(dyn, trace_estimator=estimator, n_hutchinson_samples=n_hut, regularization=True, requires_backward=True)
        f = flow.ode_func
        x = dev(g[f'{name}/x']).to(dt).requires_grad_(True)
        B = x.shape[0]
        f._eps = dev(g[f'{name}/eps'][:n_hut]).to(dt)
        a, b, c = (dev(gg[f'{name}/{k}']).to(dt) for k in 'abc')
        vel, trace, reg = f(torch.tensor(float(g[f'{name}/t'][0]), dtype=dt), (x, x.new_zeros(B), x.new_zeros(B)))
        loss = (a * vel).sum() + (b * trace).sum() + (c * reg).sum()
        key = f'{name}/{estimator}'
        np.testing.assert_allclose(float(loss.detach()), float(gg[f'{key}/loss']), rtol=max(tol, 2e-5), atol=1e-5)
        loss.backward()
        assert rel_l2(x.grad, gg[f'{key}/gx']) <= tol
        for k, p in dyn.named_parameters():
            ref = gg[f'{key}/grad/{k}']
            got = np.zeros_like(ref) if p.grad is None else p.grad.detach().cpu().numpy().astype(np.float64)
            scale = max(np.abs(ref).max(), 1e-12)
            assert np.abs(got - ref).max() / scale <= 4 * tol, (k, np.abs(got - ref).max() / scale)


def test_continuous_flow_trains():
    """A few optimiser steps of a ContinuousFlow over the EGNN dynamics (fixed-grid rk4, Hutchinson trace with fixed
    noise, the reference's regulariser in the loss): gradients reach every parameter, the loss goes down, and the
    kernels (``torch.no_grad()``) evaluate the trained flow to the same numbers as the route that trained it."""
    from tfep_amd.nn.flows import ContinuousFlow
    g = gu.load('continuous.npz')
    dyn, cfg = build_dynamics('tiny', g, split=False)
    x = dev(g['tiny/x'])
    flow = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 0.5}, n_hutchinson_samples=2, regularization=True)
    flow.ode_func.fixed_noise = dev(g['tiny/eps'][:2])
    opt = torch.optim.Adam(flow.parameters(), lr=2e-3)
    target = torch.roll(x, 3, dims=1).detach()
    losses = []
    for _ in range(6):
        opt.zero_grad()
        y, tr, reg = flow(x)
        loss = ((y - target) ** 2).sum(dim=1).mean() - tr.mean() + 0.01 * reg.mean()
        loss.backward()
        # (the node-feature update of the LAST layer does not reach the velocity: no gradient there, as in the reference)
        assert sum(p.grad is not None for p in flow.parameters()) >= len(list(flow.parameters())) - 4
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]
    y_t, tr_t, reg_t = [o.detach() for o in flow(x)]
    with torch.no_grad():
        y_k, tr_k, reg_k = flow(x)
    assert torch.allclose(y_k, y_t, rtol=1e-4, atol=1e-5) and torch.allclose(tr_k, tr_t, rtol=1e-3, atol=1e-4)
    assert torch.allclose(reg_k, reg_t, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('cutoff', [None, 0.6])
def test_split_exact_and_float64_oracle_at_64_atoms(cutoff):
    """the default (split-f16) and the exact-fp32 edge products against the float64 oracle at a size
    where every workgroup walks many sources -- 3 x 64 atoms on a jittered lattice, default widths (4 layers, 64 features,
    64 radial functions), 8 rows; all pairs inside the cutoff, and a 0.6 nm cutoff that prunes most of them.  Velocity,
    directional derivative J v and the Hutchinson term v . (J v)."""
    from tfep_amd.nn.dynamics import EGNNDynamics
    n, B = 64, 8
    gen = torch.Generator().manual_seed(11)
    side, a = 4, 0.215
    grid = torch.stack(torch.meshgrid(*[torch.arange(side, dtype=torch.float32)] * 3, indexing='ij'), -1).reshape(-1, 3)[:n]
    x = (grid[None] * a + (torch.rand(B, n, 3, generator=gen) - 0.5) * 0.3 * a).reshape(B, 3 * n)
    v = torch.randn(B, 3 * n, generator=gen)
    rc = 2.0 * side * a if cutoff is None else cutoff
    torch.manual_seed(1)
    dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=rc, initialize_identity=False)
    with torch.no_grad():
        for prm in dyn.parameters():
            prm.add_(0.05 * torch.randn(prm.shape, generator=gen))
    cfg = dict(r_cutoff=rc, time_feat_dim=16, distance_feat_dim=64, n_layers=4, speed_factor=1.0)
    sd64 = {k: (t.double() if t.is_floating_point() else t) for k, t in dyn.state_dict().items()}
    odyn = lambda t, z: oe.egnn_dynamics(t, z, sd64, cfg)
    t = torch.tensor(0.37, dtype=torch.float64)
    vel64 = odyn(t, x.double())
    jv64 = oe.jvp(odyn, t, x.double(), v.double())
    if cutoff is not None:                                  # the cutoff really prunes
        d = (x.reshape(B, n, 3)[:, :, None] - x.reshape(B, n, 3)[:, None]).norm(dim=-1)
        assert 0.05 < float(((d <= rc) & (d > 0)).float().mean()) < 0.6
    dyn = dyn.cuda()
    errs = {}
    for name, split in (('split', True), ('exact', False), ('default', None)):
        dyn.split_gemm = split
        with torch.no_grad():
            vel, jv = dyn.jvp(0.37, x.cuda(), v.cuda())
        errs[name] = (rel_l2(vel, vel64.numpy()), rel_l2(jv, jv64.numpy()))
        hut = (jv.double().cpu() * v.double()).sum(dim=1)
        np.testing.assert_allclose(hut.numpy(), (jv64 * v.double()).sum(dim=1).numpy(), rtol=2e-5, atol=2e-5 * float(jv64.abs().max()))
    print(f'64 atoms, cutoff {cutoff}: rel L2 (velocity, J v) ' + ', '.join(f'{k} {a:.1e} / {b:.1e}' for k, (a, b) in errs.items()))
    for name, (ev, ej) in errs.items():
        assert ev <= 5e-6 and ej <= 1e-5, (name, ev, ej)
    assert errs['default'] == errs['split']                 # the default arithmetic of the dynamics is the split chain


def test_grad_mode_at_cfg5_size_fails_with_the_reason_or_stays_on_the_kernels():
    """BASELINE cfg5's size under grad mode: a fresh module has trainable parameters, so ``flow(x)`` without
    ``torch.no_grad()`` takes the differentiable route -- the dynamics as DENSE torch operators, B n^2 (2F + G) values per
    layer (3.4 TB per evaluation of the dynamics here, 40 evaluations per flow).  That must fail with a message that says so (not with an allocator error), and the same call
    with frozen parameters and an input that needs no gradient must run on the kernels, grad mode or not, bit for bit like
    ``torch.no_grad()``."""
    from tfep_amd import _lib
    from tfep_amd.nn.dynamics import EGNNDynamics
    from tfep_amd.nn.flows import ContinuousFlow
    B, n = 16384, 256
    gen = torch.Generator(device='cuda').manual_seed(7)
    x = torch.randn(B, 3 * n, device='cuda', generator=gen)
    torch.manual_seed(0)
    dyn = EGNNDynamics(node_types=[i % 4 for i in range(n)], r_cutoff=4.0, speed_factor=0.02, initialize_identity=False).cuda()
    flow = ContinuousFlow(dyn, solver='rk4', solver_options={'step_size': 0.5}, regularization=False)
    assert dyn.dense_route_bytes(B) > 3 * 2 ** 40              # 3.4 TB for ONE evaluation of the dynamics (a flow keeps 40)
    with pytest.raises(_lib.TfepHipError, match='torch.no_grad'):
        flow(x)
    # (the differentiable route marks its input as requiring a gradient, in place, like the reference: continuous.py:243)
    x = x.detach().requires_grad_(False)
    small = x[:64].clone()
    flow.ode_func.fixed_noise = torch.randn(1, 64, 3 * n, device='cuda', generator=gen)
    with torch.no_grad():
        y0, t0 = flow(small)
    for p in flow.parameters():
        p.requires_grad_(False)
    y1, t1 = flow(small)                                      # grad mode on, nothing to differentiate: the kernels
    assert torch.equal(y0, y1) and torch.equal(t0, t1) and not y1.requires_grad
    flow.ode_func.fixed_noise = None
    y2, _ = flow(x[:2048])                                    # ... at a size the dense route could not hold (0.4 TB per evaluation)
    assert bool(torch.isfinite(y2).all())
