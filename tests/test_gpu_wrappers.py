"""GPU parity of the flow wrappers around the HIP MAF path (PartialFlow / CenteredCentroidFlow /
OrientedFlow, nested as TFEPMapBase nests them) against the reference run in float64
(tests/golden/wrappers.npz; properties follow tests/nn/flows/test_centroid.py and test_oriented.py)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

NAMES = list(gu.wrapper_configs())


def build(name, g):
    import tfep_amd.nn.flows as flows
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    from oracle.made import generate_degrees
    cfg = gu.wrapper_configs()[name]
    n_in = gu.wrapper_n_inner(cfg)
    if cfg.get('spline'):
        tr = NeuralSplineTransformer(x0=torch.full((n_in,), -8.0), xf=torch.full((n_in,), 8.0), n_bins=6)
    else:
        tr = AffineTransformer()
    inner = MAF(degrees_in=torch.as_tensor(generate_degrees(n_in, 'ascending')), transformer=tr,
                initialize_identity=False)
    flow = gu.build_wrapped(cfg, inner, flows)
    sd = flow.state_dict()
    for k, v in gu.sub(g, f'{name}/sd/').items():
        assert k in sd, k
        sd[k] = torch.from_numpy(np.asarray(v))
    flow.load_state_dict(sd, strict=True)
    return flow.cuda(), cfg


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('name', NAMES)
def test_wrapped_forward_matches_reference(name):
    g = gu.load('wrappers.npz')
    flow, cfg = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda()
    x0 = x.clone()
    with torch.no_grad():
        y, ldj = flow(x)
    assert torch.equal(x, x0), 'input modified'
    assert y.shape == g[f'{name}/y_f64'].shape
    noise_y = rel(g[f'{name}/y_f32'], g[f'{name}/y_f64'])
    assert rel(y.cpu(), g[f'{name}/y_f64']) < max(2 * noise_y, 2e-6)
    noise_l = np.abs(g[f'{name}/ldj_f32'].astype(np.float64) - g[f'{name}/ldj_f64']).max()
    assert np.abs(ldj.cpu().numpy().astype(np.float64) - g[f'{name}/ldj_f64']).max() < max(4 * noise_l, 2e-5)


@pytest.mark.parametrize('name', [n for n in NAMES if gu.wrapper_configs()[n]['inverse']])
def test_wrapped_inverse_matches_reference_and_round_trips(name):
    g = gu.load('wrappers.npz')
    flow, cfg = build(name, g)
    yin = torch.from_numpy(g[f'{name}/inv_in']).cuda()
    with torch.no_grad():
        x, ldj = flow.inverse(yin)
        y2, ldj2 = flow(x)
    assert rel(x.cpu(), g[f'{name}/xinv_f64']) < 2e-5
    assert np.abs(ldj.cpu().numpy().astype(np.float64) - g[f'{name}/ldjinv_f64']).max() < 1e-4
    assert torch.allclose(y2, yin, atol=2e-4)
    assert torch.allclose(ldj + ldj2, torch.zeros_like(ldj), atol=1e-3)


@pytest.mark.parametrize('name', NAMES)
def test_wrapped_gradients_match_reference_autograd(name):
    g = gu.load('wrappers.npz')
    flow, cfg = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda().requires_grad_(True)
    y, ldj = flow(x)
    c = torch.cos(torch.arange(y.shape[0], device='cuda').unsqueeze(1)
                  + 2.0 * torch.arange(y.shape[1], device='cuda').unsqueeze(0)).float()
    ((y * c).sum() + ldj.sum()).backward()
    assert rel(x.grad.cpu(), g[f'{name}/gx_f64']) < 1e-4, rel(x.grad.cpu(), g[f'{name}/gx_f64'])
    for k, p in flow.named_parameters():
        ref = g[f'{name}/gp/{k}']
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-8)
        assert err < 3e-4, (k, err)


def test_centroid_is_preserved_or_moved_to_origin():
    g = gu.load('wrappers.npz')
    flow, cfg = build('cen_default', g)
    x = torch.from_numpy(g['cen_default/x']).cuda()
    with torch.no_grad():
        y, _ = flow(x)
        assert torch.allclose(y.reshape(len(y), -1, 3).mean(1), x.reshape(len(x), -1, 3).mean(1), atol=1e-5)
        flow.translate_back = False
        y0, _ = flow(x)
        assert y0.reshape(len(y), -1, 3).mean(1).abs().max() < 1e-5
    flow, cfg = build('cen_subset_w', g)
    x = torch.from_numpy(g['cen_subset_w/x']).cuda()
    with torch.no_grad():
        y, _ = flow(x)
    w = torch.tensor([1.0, 12.0, 16.0, 14.0], device='cuda')
    w = (w / w.sum()).reshape(1, 4, 1)
    cen = (y.reshape(len(y), -1, 3)[:, [1, 3, 0, 4]] * w).sum(1)
    assert torch.allclose(cen, torch.tensor([0.5, -1.0, 2.0], device='cuda').expand_as(cen), atol=1e-5)


def test_oriented_constrained_dofs():
    g = gu.load('wrappers.npz')
    flow, cfg = build('ori_default', g)
    x = torch.from_numpy(g['ori_default/x']).cuda()
    with torch.no_grad():
        flow.rotate_back = False
        y, _ = flow(x)
        # axis point on x (y = z = 0 exactly, round-off removed), plane point on xy (z = 0)
        assert torch.all(y[:, [1, 2, 5]] == 0)
        # rotating back is a rigid motion of the same output: point-point distances agree
        flow.rotate_back = True
        yb, _ = flow(x)
        assert not torch.allclose(yb, y, atol=1e-3)
        d = torch.cdist(y.reshape(len(y), -1, 3), y.reshape(len(y), -1, 3))
        db = torch.cdist(yb.reshape(len(y), -1, 3), yb.reshape(len(y), -1, 3))
        assert torch.allclose(d, db, atol=1e-4)
    flow, cfg = build('ori_zyz_partial', g)
    with torch.no_grad():
        y, _ = flow(torch.from_numpy(g['ori_zyz_partial/x']).cuda())
    assert y.shape[1] == 12
