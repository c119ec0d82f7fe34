"""PCAWhitenedFlow around the HIP MAF path against the reference run in float64 (tests/golden/pca.npz, made by
tools/gen_golden.py:gen_pca from reference flows/pca.py), and the properties the reference tests
(tests/nn/flows/test_pca.py: identity inner flow, diagonal covariance when left whitened, round trip)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu

NAMES = list(gu.pca_configs())


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def build(name, g):
    from oracle.made import generate_degrees
    from tfep_amd.nn.flows import MAF, PCAWhitenedFlow
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    cfg = gu.pca_configs()[name]
    D = cfg['D']
    tr = NeuralSplineTransformer(x0=torch.full((D,), -9.0), xf=torch.full((D,), 9.0), n_bins=5) if cfg['spline'] \
        else AffineTransformer()
    inner = MAF(degrees_in=torch.as_tensor(generate_degrees(D, cfg['order'])), transformer=tr, initialize_identity=False)
    flow = PCAWhitenedFlow(inner, gu.pca_data(cfg), blacken=cfg['blacken'])
    own = {k: v.clone() for k, v in flow.state_dict().items()}
    sd = flow.state_dict()
    gold = gu.sub(g, f'{name}/sd/')
    assert set(gold) == {k for k in sd if not k.endswith('.mask')}
    for k, v in gold.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    flow.load_state_dict(sd, strict=True)
    return flow.cuda(), cfg, own


@pytest.mark.parametrize('name', NAMES)
def test_whitening_statistics_match_the_reference(name):
    """Eigenvectors are defined up to sign and the float32 eigh may order near-degenerate pairs differently: compare what
    the matrices DO (W W^T = cov^-1, W B = 1) and the log-det, which are unique."""
    g = gu.load('pca.npz')
    flow, cfg, own = build(name, g)
    w_ref, b_ref = g[f'{name}/sd/whitening_matrix'].astype(np.float64), g[f'{name}/sd/blackening_matrix'].astype(np.float64)
    w, b = own['whitening_matrix'].double().numpy(), own['blackening_matrix'].double().numpy()
    assert rel(w @ w.T, w_ref @ w_ref.T) < 1e-4
    assert np.abs(w @ b - np.eye(cfg['D'])).max() < 1e-4
    assert abs(float(own['whitening_log_det_J']) - float(g[f'{name}/sd/whitening_log_det_J'])) < 1e-4
    assert rel(own['mean'].numpy(), g[f'{name}/sd/mean']) < 1e-6
    assert flow.n_parameters() == int(g[f'{name}/n_parameters'])


@pytest.mark.parametrize('name', NAMES)
def test_pca_forward_and_inverse_match_reference(name):
    g = gu.load('pca.npz')
    flow, cfg, _ = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda()
    x0 = x.clone()
    with torch.no_grad():
        y, ldj = flow(x)
    assert torch.equal(x, x0), 'input modified'
    noise_y = rel(g[f'{name}/y_f32'], g[f'{name}/y_f64'])
    assert rel(y.cpu(), g[f'{name}/y_f64']) < max(2 * noise_y, 2e-6)
    noise_l = np.abs(g[f'{name}/ldj_f32'].astype(np.float64) - g[f'{name}/ldj_f64']).max()
    assert np.abs(ldj.cpu().numpy().astype(np.float64) - g[f'{name}/ldj_f64']).max() < max(4 * noise_l, 2e-5)
    yin = torch.from_numpy(g[f'{name}/inv_in']).cuda()
    with torch.no_grad():
        xi, li = flow.inverse(yin)
        y2, l2 = flow(xi)
    assert rel(xi.cpu(), g[f'{name}/xinv_f64']) < 2e-5
    assert np.abs(li.cpu().numpy().astype(np.float64) - g[f'{name}/ldjinv_f64']).max() < 1e-4
    assert torch.allclose(y2, yin, atol=5e-4) and torch.allclose(li + l2, torch.zeros_like(li), atol=1e-3)


@pytest.mark.parametrize('name', NAMES)
def test_pca_gradients_match_reference_autograd(name):
    g = gu.load('pca.npz')
    flow, cfg, _ = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda().requires_grad_(True)
    y, ldj = flow(x)
    c = torch.cos(torch.arange(y.shape[0], device='cuda').unsqueeze(1)
                  + 2.0 * torch.arange(y.shape[1], device='cuda').unsqueeze(0)).float()
    ((y * c).sum() + ldj.sum()).backward()
    assert rel(x.grad.cpu(), g[f'{name}/gx_f64']) < 1e-4
    for k, p in flow.named_parameters():
        ref = g[f'{name}/gp/{k}']
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-8)
        assert err < 3e-4, (k, err)


class _Identity(torch.nn.Module):
    def forward(self, x):
        return x, torch.zeros(len(x), device=x.device)

    def inverse(self, y):
        return self(y)

    def n_parameters(self):
        return 0


@pytest.mark.parametrize('blacken', [True, False])
def test_identity_inner_flow_properties(blacken):
    """reference tests/nn/flows/test_pca.py:58-104"""
    from tfep_amd.nn.flows import PCAWhitenedFlow
    cfg = dict(D=8, n_data=500, seed=31)
    data = gu.pca_data(cfg)
    flow = PCAWhitenedFlow(_Identity(), data, blacken=blacken).cuda()
    x = data.cuda()
    with torch.no_grad():
        y, ldj = flow(x)
        xi, li = flow.inverse(y)
    if blacken:
        assert torch.allclose(y, x, atol=2e-5) and torch.equal(ldj, torch.zeros_like(ldj))
    else:
        yc = y - y.mean(0)
        cov = (yc.t() @ yc / (len(y) - 1)).cpu()
        assert torch.allclose(cov, torch.eye(8), atol=2e-4) and float(y.mean(0).abs().max()) < 1e-5
        assert float(ldj.abs().min()) > 1e-3 and torch.allclose(ldj, ldj[0].expand_as(ldj))
    assert torch.allclose(xi, x, atol=5e-5) and torch.allclose(ldj + li, torch.zeros_like(ldj), atol=1e-6)


def test_errors_and_state():
    from tfep_amd.nn.flows import PCAWhitenedFlow
    data = gu.pca_data(dict(D=5, n_data=200, seed=5))
    flow = PCAWhitenedFlow(_Identity(), data)
    assert list(flow.state_dict()) == ['mean', 'whitening_matrix', 'blackening_matrix', 'whitening_log_det_J']
    with pytest.raises(ValueError):
        PCAWhitenedFlow(_Identity(), data[0])
    with pytest.raises(Exception):                       # CPU tensors never reach a kernel
        flow(data)
    flow = flow.cuda()
    with pytest.raises(RuntimeError, match='features'):
        flow(torch.zeros(4, 6, device='cuda'))
    # buffers replaced by load_state_dict are picked up by the kernels' float32 copies
    with torch.no_grad():
        y1, _ = flow(data[:7].cuda())
        sd = flow.state_dict()
        sd['mean'] = sd['mean'] + 1.0
        flow.load_state_dict(sd)
        y2, _ = flow(data[:7].cuda())
    assert torch.allclose(y1, y2, atol=1e-5)             # (blackening adds the mean back)
    flow.blacken = False
    with torch.no_grad():
        y3, _ = flow(data[:7].cuda())
        sd['mean'] = sd['mean'] - 1.0
        flow.load_state_dict(sd)
        y4, _ = flow(data[:7].cuda())
    assert not torch.allclose(y3, y4, atol=1e-3)
