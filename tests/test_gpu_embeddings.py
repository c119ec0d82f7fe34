"""GPU tests of the embeddings in front of the MADE conditioner: module outputs, and MAF layers with a
flip-invariant / mixed embedding -- forward, inverse and gradients -- against the reference in float64
(tests/golden/embeddings.npz)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('name', list(gu.embedding_configs()))
def test_embedding_module_on_device(name):
    import tfep_amd.nn.embeddings as E
    g = gu.load('embeddings.npz')
    cfg = gu.embedding_configs()[name]
    emb = gu.build_embedding(cfg, E)
    emb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gu.sub(g, f'emb/{name}/sd/').items()})
    emb = emb.cuda()
    x = torch.from_numpy(g[f'emb/{name}/x']).cuda()
    with torch.no_grad():
        out = emb(x)
    assert rel(out.cpu(), g[f'emb/{name}/out_f64']) < 2e-6
    # differentiable on the device (mixed: through the HIP periodic-embedding backward)
    xg = x.clone().requires_grad_(True)
    emb(xg).square().sum().backward()
    assert torch.isfinite(xg.grad).all() and float(xg.grad.abs().max()) > 0


def build(name, g):
    import tfep_amd.nn.embeddings as E
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    cfg = gu.embedded_flow_configs()[name]
    n_tr = sum(1 for d in cfg['degrees_in'] if d >= 0)
    tr = (NeuralSplineTransformer(x0=torch.full((n_tr,), -4.0), xf=torch.full((n_tr,), 4.0), n_bins=8)
          if cfg['transformer'] == 'spline' else AffineTransformer())
    flow = SequentialFlow(MAF(degrees_in=torch.as_tensor(cfg['degrees_in']), transformer=tr,
                              embedding=gu.build_embedding(cfg['embedding'], E), initialize_identity=False))
    sd = flow.state_dict()
    gold = gu.sub(g, f'{name}/sd/')
    assert {k for k in sd if not k.endswith('.mask')} == set(gold), {k for k in sd if not k.endswith('.mask')} ^ set(gold)
    for k, v in gold.items():
        sd[k] = torch.from_numpy(np.asarray(v))
    flow.load_state_dict(sd, strict=True)
    return flow.cuda()


@pytest.mark.parametrize('name', list(gu.embedded_flow_configs()))
def test_embedded_flow_forward_inverse(name):
    g = gu.load('embeddings.npz')
    flow = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda()
    with torch.no_grad():
        y, ldj = flow(x)
        xi, li = flow.inverse(torch.from_numpy(g[f'{name}/inv_in']).cuda())
    assert rel(y.cpu(), g[f'{name}/y_f64']) < max(2 * rel(g[f'{name}/y_f32'], g[f'{name}/y_f64']), 2e-6)
    noise = np.abs(g[f'{name}/ldj_f32'].astype(np.float64) - g[f'{name}/ldj_f64']).max()
    assert np.abs(ldj.cpu().numpy().astype(np.float64) - g[f'{name}/ldj_f64']).max() < max(4 * noise, 2e-5)
    assert rel(xi.cpu(), g[f'{name}/xinv_f64']) < 2e-5
    assert np.abs(li.cpu().numpy().astype(np.float64) - g[f'{name}/ldjinv_f64']).max() < 1e-4


@pytest.mark.parametrize('name', list(gu.embedded_flow_configs()))
def test_embedded_flow_gradients(name):
    """Gradients of x, of the conditioner AND of the embedding networks against the reference's autograd."""
    g = gu.load('embeddings.npz')
    flow = build(name, g)
    x = torch.from_numpy(g[f'{name}/x']).cuda().requires_grad_(True)
    y, ldj = flow(x)
    c = torch.cos(torch.arange(y.shape[0], device='cuda').unsqueeze(1)
                  + 2.0 * torch.arange(y.shape[1], device='cuda').unsqueeze(0)).float()
    ((y * c).sum() + ldj.sum()).backward()
    assert rel(x.grad.cpu(), g[f'{name}/gx_f64']) < 1e-4
    seen = 0
    # the last bias of the weight network has an exactly zero gradient (softmax over the pair is shift invariant):
    # errors are measured against the larger of the entry scale and 1e-4 of the largest gradient in the model
    gmax = max(np.abs(g[f'{name}/gp/{k}']).max() for k, _ in flow.named_parameters())
    for k, p in flow.named_parameters():
        ref = g[f'{name}/gp/{k}']
        assert p.grad is not None, k
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / max(np.abs(ref).max(), 1e-4 * gmax)
        assert err < 3e-4, (k, err)
        seen += 'embedding_layer' in k or 'weight_layer' in k
    assert seen >= 8                                   # the two small networks of the flip-invariant embedding
