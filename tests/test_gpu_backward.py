"""GPU tests of the training step: gradients of loss = BoltzmannKLDivLoss(u_B(y), log_det_J) through the
HIP backward kernels against the reference's own autograd run in float64 (tests/golden/grads.npz)."""
import numpy as np
import pytest
import torch

import golden_util as gu

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def test_masked_linear_function_gradients():
    """MaskedLinearFunc.backward semantics (masked.py:279-302) through tfep_masked_linear_gemm."""
    from tfep_amd.nn.flows import _backward as bw
    from tfep_amd import ops
    g = gu.load('grads.npz')
    x, w, b, m, gy = (torch.from_numpy(g[f'ml/{k}']).float().cuda() for k in ('x', 'w', 'b', 'mask', 'gy'))
    tm, tn, tk = ops.tile_sizes()
    B, K = x.shape
    N = w.shape[0]
    kp, npad = ops.round_up(K, tk), ops.round_up(N, tk)
    wp = ops.masked_weight_prepare(w, None, m, n_rows_padded=npad, k_padded=kp)
    f32 = dict(dtype=torch.float32, device='cuda')
    # grad_input = g (W o M)
    wt = bw._transpose(wp, npad, kp, torch.zeros(kp, npad, **f32))
    gpad = ops.pad_columns(gy, npad)
    gx = bw._gemm(gpad, wt, torch.empty(B, kp, **f32), B, kp, kp)
    assert rel(gx[:, :K].cpu(), g['ml/gx']) < 1e-6
    # grad_weight = (g^T x) o M, grad_bias = column sums
    Bp = ops.round_up(B, tk)
    gT = bw._transpose(gpad, B, npad, torch.zeros(npad, Bp, **f32))
    xT = bw._transpose(ops.pad_columns(x, kp), B, kp, torch.zeros(kp, Bp, **f32))
    gw = bw._gemm(gT, xT, torch.zeros(npad, kp, **f32), npad, kp, kp, accumulate=1)
    gw = gw[:N, :K] * m
    assert rel(gw.cpu(), g['ml/gw']) < 1e-6


@pytest.mark.parametrize('split', [False, True])          # exact-fp32 MFMA GEMMs / split-f16 GEMMs (forced: these are small)
@pytest.mark.parametrize('name', ['affine', 'spline', 'circular', 'identslopes', 'moebius', 'mixed', 'learnlow', 'learnup', 'learnboth'])
def test_training_step_gradients_match_reference_autograd(name, split):
    from tfep_amd.loss import BoltzmannKLDivLoss
    g = gu.load('grads.npz')
    flow = gu.build_flow(name, g, configs=gu.grad_flow_configs())
    for layer in flow:
        layer.split_gemm = split
    x = torch.from_numpy(g[f'{name}/x']).cuda().requires_grad_(True)
    c, d = torch.from_numpy(g[f'{name}/c']).cuda(), torch.from_numpy(g[f'{name}/d']).cuda()
    y, ldj = flow(x)
    assert y.requires_grad and ldj.requires_grad
    u_b = (c * y ** 2 + d * y).sum(dim=1)            # the (external) target potential: plain torch
    loss = BoltzmannKLDivLoss()(u_b, ldj)
    np.testing.assert_allclose(float(loss.detach()), float(g[f'{name}/loss_f64']), rtol=2e-5)
    loss.backward()
    assert rel(x.grad.cpu(), g[f'{name}/gx_f64']) < 5e-5, rel(x.grad.cpu(), g[f'{name}/gx_f64'])
    worst = 0.0
    for k, p in flow.named_parameters():
        ref = g[f'{name}/grad/{k}']
        assert p.grad is not None and tuple(p.grad.shape) == ref.shape, k
        scale = max(np.abs(ref).max(), 1e-8)
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / scale
        worst = max(worst, err)
        assert err < 2e-4, (k, err)
    # masked weights never receive gradient (masked.py:401-402)
    for layer in flow:
        for lin in layer._conditioner.layers[::2]:
            wv = lin.weight_v if lin.has_weight_norm else lin._parameters['weight']
            assert torch.all(wv.grad[lin.mask == 0] == 0)


@pytest.mark.parametrize('name', ['affine', 'spline', 'circular', 'identslopes', 'moebius', 'mixed', 'learnlow', 'learnup', 'learnboth'])
def test_inverse_is_differentiable_like_the_reference(name):
    """``flow.inverse`` under autograd (reference autoregressive.py:179-229 is plain differentiable torch): values, the
    gradient of y and of EVERY parameter against the reference's float64 autograd through its own inverse
    (tests/golden/inv_grads.npz: loss = KL(u(x), log_det_J_inv) with a quadratic u).  The blocked inverse (no graph) must
    return the same numbers as the differentiable route."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    g = gu.load('inv_grads.npz')
    flow = gu.build_flow(name, g, configs=gu.grad_flow_configs())
    y = torch.from_numpy(g[f'{name}/y']).cuda().requires_grad_(True)
    c, d = torch.from_numpy(g[f'{name}/c']).cuda(), torch.from_numpy(g[f'{name}/d']).cuda()
    x, ldj = flow.inverse(y)
    assert x.requires_grad and ldj.requires_grad
    assert rel(x.detach().cpu(), g[f'{name}/x_f64']) < 2e-5
    np.testing.assert_allclose(ldj.detach().cpu().numpy(), g[f'{name}/ldj_f64'], rtol=2e-5, atol=2e-5)
    with torch.no_grad():
        xb, lb = flow.inverse(y.detach())
    assert torch.equal(xb, x.detach()) and torch.equal(lb, ldj.detach())          # values: the fast path either way
    loss = BoltzmannKLDivLoss()((c * x ** 2 + d * x).sum(dim=1), ldj)
    np.testing.assert_allclose(float(loss.detach()), float(g[f'{name}/loss_f64']), rtol=2e-5)
    loss.backward()
    assert rel(y.grad.cpu(), g[f'{name}/gy_f64']) < 1e-4, rel(y.grad.cpu(), g[f'{name}/gy_f64'])
    for k, p in flow.named_parameters():
        ref = g[f'{name}/grad/{k}']
        assert p.grad is not None and tuple(p.grad.shape) == ref.shape, k
        scale = max(np.abs(ref).max(), 1e-8)
        err = np.abs(p.grad.cpu().numpy().astype(np.float64) - ref).max() / scale
        assert err < 3e-4, (k, err)


def _wide_flow():
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, SequentialFlow
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D = 300
    flow = SequentialFlow(*[MAF(generate_degrees(D, o), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                                initialize_identity=False) for o in ('ascending', 'descending')])
    gu.wide_parameters(flow, 400)
    return flow.cuda()


@pytest.mark.parametrize('arithmetic', ['exact', 'split', 'default'])
def test_wide_layer_gradients_and_adamw_step_against_reference(arithmetic):
    """a 2-layer RQ-8 MAF at D = 300 with the default hidden width (1498; 13.9 M weights per layer) -- the
    size at which the backward's k-ranges, prefix packs and split transposes are multi-tile -- against the reference's
    float64 autograd (tests/golden/grads_wide.npz: whole small tensors, 4096 sampled entries of the big ones, the
    reference's own float32 results beside them).  Per tensor: relative L2 <= max(1e-5, 4 x the reference's float32
    noise), and COMPONENT-WISE every entry above 1e-4 of the tensor's maximum within 2e-3 relative (the reference's float32
    entries are held to the same bound: it is what fp32 accumulation over 256 samples gives); then one AdamW step from the
    same state: the parameter deltas (Adam normalises every entry by its own magnitude) at the same entries."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    g = gu.load('grads_wide.npz')
    flow = _wide_flow()
    for layer in flow:
        layer.split_gemm = {'exact': False, 'split': True, 'default': None}[arithmetic]
    x = torch.from_numpy(g['x']).cuda().requires_grad_(True)
    c, d = torch.from_numpy(g['c']).cuda(), torch.from_numpy(g['d']).cuda()
    before = {k: p.detach().clone() for k, p in flow.named_parameters()}
    y, ldj = flow(x)
    loss = BoltzmannKLDivLoss()((c * y ** 2 + d * y).sum(dim=1), ldj)
    np.testing.assert_allclose(float(loss.detach()), float(g['loss_f64']), rtol=1e-5)
    loss.backward()
    assert rel(x.grad.cpu(), g['gx_f64']) < max(1e-5, 4 * rel(g['gx_f32'], g['gx_f64']))
    report = {}
    for k, p in flow.named_parameters():
        ours = p.grad.detach().double().cpu().numpy()
        if f'full/{k}' in g:
            ref, ref32, ours_s = g[f'full/{k}'], g[f'full32/{k}'], ours
        else:
            idx = g[f'idx/{k}']
            ref, ref32, ours_s = g[f'val/{k}'], g[f'val32/{k}'], ours.reshape(-1)[idx]
            # the whole tensor's norm against the reference's (the samples cannot see a wrong tile elsewhere)
            assert abs(np.linalg.norm(ours) / float(g[f'norm/{k}']) - 1.0) < 1e-5, k
        noise = float(g[f'noise/{k}'])
        r = np.linalg.norm(ours_s - ref) / np.linalg.norm(ref)
        mx = float(g[f'max/{k}'])
        row = [r, noise]
        for thr in (1e-2, 1e-4):                       # entries above 1e-2 / 1e-4 of the tensor's largest gradient
            big = np.abs(ref) > thr * mx
            row += [float((np.abs(ours_s - ref)[big] / np.abs(ref)[big]).max()), float((np.abs(ref32 - ref)[big] / np.abs(ref)[big]).max())]
        report[k] = row
    for k, (r, noise, cw2, cw2_32, cw4, cw4_32) in report.items():
        print(f'{arithmetic:8s} {k:38s} rel L2 {r:.2e} (ref fp32 {noise:.2e})  component-wise > 1e-2 max: {cw2:.2e} (ref fp32 {cw2_32:.2e})'
              f'  > 1e-4 max: {cw4:.2e} (ref fp32 {cw4_32:.2e})')
    for k, (r, noise, cw2, cw2_32, cw4, cw4_32) in report.items():
        assert r <= max(1e-5, 4 * noise), (k, r, noise)
        # component-wise: an entry is a sum over 256 samples with cancellation, so its error scales with sum |g x|, not with
        # the entry; the reference's own float32 run is held to the same bounds (its worst entries are beside ours)
        assert cw2 <= max(5e-4, 8 * cw2_32), (k, cw2, cw2_32)
        assert cw4 <= max(2e-2, 8 * cw4_32), (k, cw4, cw4_32)
    # one AdamW step (lr 1e-3, weight decay 0.01: the golden's) from the same state
    opt = torch.optim.AdamW(flow.parameters(), lr=1e-3, weight_decay=0.01)
    opt.step()
    for k, p in flow.named_parameters():
        delta = (p.detach() - before[k]).double().cpu().numpy().reshape(-1)
        if f'full/{k}' in g:
            ref_d, gref = g[f'adamw/{k}'].reshape(-1), g[f'full/{k}'].reshape(-1)
        else:
            delta, ref_d, gref = delta[g[f'idx/{k}']], g[f'adamw/{k}'], g[f'val/{k}']
        # where the gradient is far above Adam's epsilon the step is -lr (sign(g) + wd p) to rounding: compare per entry
        # (masked entries have exactly zero gradient on both sides: their step is the weight decay alone)
        clear = (np.abs(gref) > 1e-6) | (gref == 0)
        assert clear.mean() > 0.9, k
        err = np.abs(delta - ref_d)[clear].max()
        assert err < 2e-3 * 1e-3, (k, err)          # 0.2 % of the step length lr


def test_optimizer_step_runs_and_lowers_the_loss():
    """A few AdamW steps on a fixed batch (the loop of TFEPMapBase.training_step / configure_optimizers)."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    g = gu.load('grads.npz')
    flow = gu.build_flow('spline', g, configs=gu.grad_flow_configs())
    x = torch.from_numpy(g['spline/x']).cuda()
    c, d = torch.from_numpy(g['spline/c']).cuda(), torch.from_numpy(g['spline/d']).cuda()
    opt = torch.optim.AdamW(flow.parameters(), lr=5e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        y, ldj = flow(x)
        loss = BoltzmannKLDivLoss()((c * y ** 2 + d * y).sum(dim=1), ldj)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


def test_user_supplied_conditioner_trains_through_the_transformer_vjp():
    """A conditioner that is just a torch module (no HIP backward of its own): autograd differentiates it, the transformer
    goes through its VJP kernel (``generic_forward``).  Gradients of x and of the conditioner's parameters against the
    same map written in float64 torch (affine.py:321-323: y = x exp(log_scale) + shift, log|det J| = sum log_scale)."""
    from tfep_amd.nn.conditioners.conditioner import Conditioner
    from tfep_amd.nn.flows import AutoregressiveFlow
    from tfep_amd.nn.transformers import AffineTransformer

    class MLP(Conditioner):
        def __init__(self, d):
            super().__init__()
            self.net = torch.nn.Sequential(torch.nn.Linear(d, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2 * d))

        def forward(self, x):
            return self.net(x)

        def set_output(self, output):
            self.net[-1].bias.data = output

    torch.manual_seed(2)
    D, B = 6, 37
    flow = AutoregressiveFlow(D, [[i] for i in range(D)], MLP(D), AffineTransformer()).cuda()
    x = torch.randn(B, D, device='cuda', requires_grad=True)
    c = torch.linspace(0.5, 1.5, D, device='cuda')
    y, ldj = flow(x)
    ((c * y ** 2).sum() + 0.3 * ldj.sum()).backward()
    got = [x.grad] + [p.grad for p in flow.parameters()]
    # float64 reference
    ref_net = torch.nn.Sequential(torch.nn.Linear(D, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2 * D)).double().cuda()
    ref_net.load_state_dict({k: v.double() for k, v in flow._conditioner.net.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    th = ref_net(xr).reshape(B, 2, D)
    yr = xr * torch.exp(th[:, 1]) + th[:, 0]
    ((c.double() * yr ** 2).sum() + 0.3 * th[:, 1].sum()).backward()
    ref = [xr.grad] + [p.grad for p in ref_net.parameters()]
    assert rel(y.detach().cpu(), yr.detach().cpu()) < 1e-6
    for a, b in zip(got, ref):
        assert a is not None and rel(a.cpu(), b.cpu()) < 2e-5


@pytest.mark.parametrize('name', ['spline', 'moebius', 'circular'])
def test_generic_autograd_path_matches_the_fused_backward(name):
    """The same MADE layer differentiated two ways: the one-node HIP backward (``MAFLayerFunction``) and the generic path
    (MaskedLinear autograd + transformer VJP) that layers outside ``supported`` take."""
    from tfep_amd.nn.flows import _backward as bw
    g = gu.load('grads.npz')
    flow = gu.build_flow(name, g, configs=gu.grad_flow_configs())
    layer = flow[0]
    x0 = torch.from_numpy(g[f'{name}/x']).float().cuda()
    c = torch.linspace(0.2, 0.9, x0.shape[1], device='cuda')

    def grads(generic):
        for p in layer.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y, l = bw.generic_forward(layer, x) if generic else layer(x)
        ((c * y ** 2).sum() + 0.5 * l.sum()).backward()
        return [x.grad.clone()] + [p.grad.clone() for p in layer.parameters() if p.grad is not None]

    a, b = grads(False), grads(True)
    assert len(a) == len(b)
    for u, v in zip(a, b):
        scale = float(u.abs().max()) + 1e-9
        assert float((u - v).abs().max()) <= 2e-4 * scale + 1e-6


def test_user_supplied_torch_transformer_and_conditioner_train_by_autograd():
    """any user-supplied transformer / conditioner must still work.  Both written in torch: the layer runs the
    unfused path and autograd differentiates everything; against the same model in float64."""
    from tfep_amd.nn.conditioners.conditioner import Conditioner
    from tfep_amd.nn.flows import AutoregressiveFlow
    from tfep_amd.nn.transformers.transformer import Transformer

    class Lin(Conditioner):
        def __init__(self, d):
            super().__init__()
            self.lin = torch.nn.Linear(d, d)

        def forward(self, x):
            return self.lin(x)

        def set_output(self, output):
            self.lin.bias.data = output

    class Shift(Transformer):                       # y = x + tanh(theta), volume preserving
        def forward(self, x, parameters):
            return x + torch.tanh(parameters), torch.zeros(x.shape[0], device=x.device)

        def inverse(self, y, parameters):
            return y - torch.tanh(parameters), torch.zeros(y.shape[0], device=y.device)

        def get_identity_parameters(self, n_features):
            return torch.zeros(n_features)

        def get_degrees_out(self, degrees_in):
            return degrees_in

    torch.manual_seed(4)
    D, B = 5, 21
    flow = AutoregressiveFlow(D, [[i] for i in range(D)], Lin(D), Shift(), initialize_identity=False).cuda()
    x = torch.randn(B, D, device='cuda', requires_grad=True)
    y, ldj = flow(x)
    (y ** 3).sum().backward()
    ref = torch.nn.Linear(D, D).double().cuda()
    ref.load_state_dict({k: v.double() for k, v in flow._conditioner.lin.state_dict().items()})
    xr = x.detach().double().requires_grad_(True)
    ((xr + torch.tanh(ref(xr))) ** 3).sum().backward()
    assert rel(x.grad.cpu(), xr.grad.cpu()) < 1e-5
    for p, q in zip(flow._conditioner.lin.parameters(), ref.parameters()):
        assert rel(p.grad.cpu(), q.grad.cpu()) < 1e-5


def test_unsupported_backward_fails_loudly():
    """A transformer without a VJP kernel: the forward works, .backward() raises."""
    from tfep_amd.nn.conditioners.conditioner import Conditioner
    from tfep_amd.nn.flows import AutoregressiveFlow
    from tfep_amd.nn.transformers import AffineTransformer

    class Constant(Conditioner):
        def __init__(self):
            super().__init__()
            self.out = torch.nn.Parameter(torch.zeros(8))

        def forward(self, x):
            return self.out.expand(x.shape[0], -1).contiguous()

        def set_output(self, output):
            self.out.data = output

    class MyAffine(AffineTransformer):          # a subclass may change the map: it has no kernel of its own
        pass

    flow = AutoregressiveFlow(4, [[0, 1], [2, 3]], Constant(), MyAffine()).cuda()
    y, ldj = flow(torch.randn(3, 4, device='cuda'))
    with pytest.raises(NotImplementedError, match='backward needs a transformer with a VJP kernel'):
        (y.sum() + ldj.sum()).backward()


@pytest.mark.parametrize('order,cond', [('ascending', None), ('descending', [1, 4])])
def test_autoregressive_property_by_gradient_sparsity(order, cond):
    """The reference's check_autoregressive_property (tests/nn/__init__.py:25-96; used by test_made.py and
    test_maf.py:288-295): back-propagate each output-degree group and require EXACT zeros in the input
    gradient where the autoregressive structure forbids a dependence, non-zeros where it allows one."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    D = 9
    deg = generate_degrees(D, order, conditioning_indices=cond)
    n_tr = int((deg != -1).sum())
    torch.manual_seed(3)
    maf = MAF(deg, transformer=NeuralSplineTransformer(torch.full((n_tr,), -3.0), torch.full((n_tr,), 3.0), 8),
              initialize_identity=False).cuda()
    x = (torch.randn(1, D, generator=torch.Generator().manual_seed(4)) * 0.8).cuda().requires_grad_(True)
    y, ldj = maf(x)
    deg_l = deg.tolist()
    for d_out in sorted(set(deg_l) - {-1}):
        idx = [i for i, d in enumerate(deg_l) if d == d_out]
        (g,) = torch.autograd.grad(y[0, idx].sum(), x, retain_graph=True)
        g = g[0].cpu()
        for j, d_in in enumerate(deg_l):
            if d_in == -1:
                assert g[j] != 0, (d_out, j)                    # conditioning features reach every output
            elif d_in < d_out or j in idx:
                assert g[j] != 0, (d_out, j)                    # allowed dependence (and y_i on x_i itself)
            else:
                assert g[j] == 0, (d_out, j, float(g[j]))       # forbidden: exactly zero
    # fixed (conditioning) features pass through unchanged
    if cond:
        assert torch.equal(y[0, cond], x[0, cond])


def test_masked_linear_module_and_functional_autograd():
    """tests/nn/test_masked.py:150-218 in spirit: the functional form matches the reference's analytic
    gradients (float64 golden), and a weight-normalised MaskedLinear trains without NaNs while masked
    weights stay exactly zero after an SGD step, including a fully-masked row."""
    from tfep_amd.nn import masked
    g = gu.load('grads.npz')
    x, w, b, m, gy = (torch.from_numpy(g[f'ml/{k}']).float().cuda() for k in ('x', 'w', 'b', 'mask', 'gy'))
    x.requires_grad_(True); w.requires_grad_(True); b.requires_grad_(True)
    y = masked.masked_linear(x, w, b, m)
    y.backward(gy)
    assert rel(x.grad.cpu(), g['ml/gx']) < 1e-6 and rel(w.grad.cpu(), g['ml/gw']) < 1e-6
    assert rel(b.grad.cpu(), g['ml/gb']) < 1e-6

    mask = torch.tril(torch.ones(5, 8))
    mask[2] = 0.0
    torch.manual_seed(0)
    lin = masked.masked_weight_norm(masked.MaskedLinear(8, 5, mask=mask)).cuda()
    ref_w = lin.weight.clone()
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    xin = torch.randn(16, 8, device='cuda')
    loss = (lin(xin) ** 2).sum()
    loss.backward()
    assert torch.all(lin.weight_v.grad[mask.cuda() == 0] == 0) and lin.weight_g.grad[2] == 0
    assert torch.isfinite(lin.weight_v.grad).all() and torch.isfinite(lin.weight_g.grad).all()
    opt.step()
    w2 = lin.weight
    assert torch.isfinite(w2).all() and torch.all(w2[mask.cuda() == 0] == 0) and not torch.equal(w2, ref_w)
    # weight-norm gradients against autograd through the plain formula (float64 on the host)
    v = lin.weight_v.detach().double().cpu().requires_grad_(True)
    gg = lin.weight_g.detach().double().cpu().requires_grad_(True)
    norm = v.norm(dim=1, keepdim=True)
    weff = torch.where(mask.double() == 0, torch.zeros_like(v), v * (gg / norm))
    lin2 = masked.masked_weight_norm(masked.MaskedLinear(8, 5, mask=mask)).cuda()
    lin2.load_state_dict(lin.state_dict())
    (lin2(xin) ** 2).sum().backward()
    ((xin.double().cpu() @ weff.t() + lin.bias.detach().double().cpu()) ** 2).sum().backward()
    live = mask != 0
    assert rel(lin2.weight_v.grad.cpu()[live], v.grad[live]) < 1e-5
    rows = [0, 1, 3, 4]
    assert rel(lin2.weight_g.grad.cpu()[rows], gg.grad[rows]) < 1e-5


def test_partial_flow_semantics_and_gradients():
    """tests/nn/flows/test_partial.py in spirit: fixed DOFs are untouched and invisible to the wrapped flow,
    the others equal the wrapped flow on the sub-vector; inverse round-trips; return_partial; gradients
    flow to the mapped inputs through the wrapped flow and to the fixed inputs as the identity."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, PartialFlow, SequentialFlow
    torch.manual_seed(0)
    D, fixed = 9, [1, 4, 8]
    prop = [i for i in range(D) if i not in fixed]
    inner = SequentialFlow(MAF(generate_degrees(len(prop)), initialize_identity=False),
                           MAF(generate_degrees(len(prop), 'descending'), initialize_identity=False))
    pf = PartialFlow(inner, fixed_indices=fixed).cuda()
    assert int(pf.n_parameters()) == int(inner.n_parameters())
    x = torch.randn(13, D, device='cuda')
    with torch.no_grad():
        y, l = pf(x)
        y_in, l_in = inner(x[:, prop].contiguous())
        assert torch.equal(y[:, fixed], x[:, fixed]) and torch.equal(y[:, prop], y_in) and torch.equal(l, l_in)
        xi, li = pf.inverse(y)
        assert torch.allclose(xi, x, atol=1e-5) and torch.allclose(l + li, torch.zeros_like(l), atol=1e-4)
        pf.return_partial = True
        yp, lp = pf(x)
        assert torch.equal(yp, y_in)
        pf.return_partial = False
    xg = x.clone().requires_grad_(True)
    y, l = pf(xg)
    (y.sum() + l.sum()).backward()
    xs = x[:, prop].contiguous().requires_grad_(True)
    ys, ls = inner(xs)
    (ys.sum() + ls.sum()).backward()
    assert torch.allclose(xg.grad[:, prop], xs.grad, atol=1e-5)
    assert torch.equal(xg.grad[:, fixed], torch.ones(13, len(fixed), device='cuda'))
    # no fixed indices: a transparent wrapper
    pf0 = PartialFlow(inner, fixed_indices=[]).cuda()
    with torch.no_grad():
        x6 = torch.randn(5, len(prop), device='cuda')
        assert torch.equal(pf0(x6)[0], inner(x6)[0])


@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('embedding', [False, True])
def test_backward_in_several_batch_chunks_equals_one_chunk(monkeypatch, split, embedding):
    """The backward processes the batch in chunks sized by ``_CHUNK_BYTES`` (one chunk for everything the other tests
    run): force 4 ragged chunks and compare every gradient with the one-chunk result; gradients accumulate over chunks in
    a fixed order, so two runs agree bit for bit."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF, _backward as bw
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    from tfep_amd import ops
    torch.manual_seed(11)
    D, B = 70, 900
    emb = PeriodicEmbedding(D, limits=[0.0, 2.0], periodic_indices=list(range(0, D, 3))) if embedding else None
    lo, hi = (0.0, 2.0) if embedding else (-4.0, 4.0)
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), lo), torch.full((D,), hi), 8,
                                                                                   circular=embedding),
              embedding=emb, hidden_layers=[150, 130], initialize_identity=False).cuda()
    maf.split_gemm = split
    x0 = (torch.rand(B, D) * 2.0 if embedding else torch.randn(B, D) * 1.2).cuda()
    c = torch.linspace(0.1, 0.4, D, device='cuda')

    def grads():
        for p in maf.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y, l = maf(x)
        BoltzmannKLDivLoss()((c * y ** 2).sum(dim=1), l).backward()
        return [x.grad.clone()] + [p.grad.clone() for p in maf.parameters()]

    one = grads()
    tm = ops.tile_sizes()[0]
    n_out_pad = ops.round_up(25 * D, ops.tile_sizes()[2])
    monkeypatch.setattr(bw, '_CHUNK_BYTES', 4 * n_out_pad * tm)          # chunk = one row tile: ceil(900 / tm) chunks
    assert B > 3 * tm
    many = grads()
    again = grads()
    names = ['x'] + [n for n, _ in maf.named_parameters()]
    for n, a, b, b2 in zip(names, one, many, again):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b).abs().max()) <= 2e-5 * scale + 1e-7, n
        assert torch.equal(b, b2), (n, float((b - b2).abs().max()), int(((b - b2) != 0).sum()),
                                    torch.nonzero((b - b2) != 0)[:6].tolist())


@pytest.mark.parametrize('split', [False, True])
@pytest.mark.parametrize('kind', ['spline', 'spline4bins', 'circular', 'affine', 'fixed+periodic', 'fixed+periodic+5bins'])
def test_activation_saving_forward_equals_the_recomputing_one(monkeypatch, split, kind):
    """A training forward keeps the hidden activations and transformer parameters for its backward when they fit
    ``_SAVE_BYTES`` (un-fused kernels on the backward's weight packing); otherwise the backward recomputes them.  Same
    kernels on the same operands either way: y / log|det J| agree with the fused forward to rounding and so does
    every gradient; the packed weights are shared between the two passes and re-made after an optimiser step."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF, _backward as bw
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    torch.manual_seed(5)
    D, B = 66, 2200                    # (above the 1 MiB of transformer parameters from which activations are kept)
    if kind == 'affine':
        deg, emb, tr = generate_degrees(D, 'descending'), None, AffineTransformer()
    elif kind in ('spline', 'spline4bins', 'circular'):
        deg, emb = generate_degrees(D, 'ascending'), None
        tr = NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 4 if kind == 'spline4bins' else 8,
                                     circular=kind == 'circular')
    else:
        fixed = [3, 17, 40]
        deg = generate_degrees(D, 'ascending', conditioning_indices=fixed)
        emb = PeriodicEmbedding(D, limits=[-4.0, 4.0], periodic_indices=list(range(1, D, 4)))
        # (5 bins: one of the fused layouts -- the saving forward is then ONE launch of the fused kernel; 6: un-fused)
        tr = NeuralSplineTransformer(torch.full((D - 3,), -4.0), torch.full((D - 3,), 4.0), 5 if kind.endswith('5bins') else 6)
    maf = MAF(deg, transformer=tr, embedding=emb, hidden_layers=[140, 170], initialize_identity=False).cuda()
    maf.split_gemm = split
    x0 = (torch.randn(B, D) * 1.3).clamp(-3.9, 3.9).cuda()
    c = torch.linspace(0.1, 0.4, D, device='cuda')

    def step():
        for p in maf.parameters():
            p.grad = None
        x = x0.clone().requires_grad_(True)
        y, l = maf(x)
        BoltzmannKLDivLoss()((c * y ** 2).sum(dim=1), l).backward()
        return y.detach(), l.detach(), [x.grad.clone()] + [p.grad.clone() for p in maf.parameters()]

    assert bw.saves_activations(maf, x0)
    ys, ls, gs = step()
    assert ('bwd_weights', str(x0.device)) in maf._dev
    # split-f16 operands + a fused spline layout: output layer, spline and parameter store in one launch on the backward's
    # feature-major packing (tfep_fused_output_transformer_forward_split_saving); same numbers as the un-fused kernels
    fused_saving = maf._dev.get(('fused_saving', str(x0.device)))
    assert (fused_saving is not None) == (split and kind in ('spline', 'spline4bins', 'circular', 'fixed+periodic+5bins'))
    if fused_saving is not None:
        monkeypatch.setenv('TFEP_FUSED_SAVING', '0')
        maf._dev.pop(('fused_saving', str(x0.device)))
        yu, lu, gu = step()
        assert maf._dev[('fused_saving', str(x0.device))] is None
        monkeypatch.delenv('TFEP_FUSED_SAVING')
        maf._dev.pop(('fused_saving', str(x0.device)))
        assert float((ys - yu).abs().max()) < 2e-5 and float((ls - lu).abs().max()) < 2e-4
        for n, a, b in zip(['x'] + [n for n, _ in maf.named_parameters()], gs, gu):
            assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-8, n
    monkeypatch.setattr(bw, '_SAVE_BYTES', 0)
    assert not bw.saves_activations(maf, x0)
    yr, lr, gr = step()
    monkeypatch.undo()
    names = ['x'] + [n for n, _ in maf.named_parameters()]
    assert float((ys - yr).abs().max()) < 2e-5 and float((ls - lr).abs().max()) < 2e-4
    for n, a, b in zip(names, gs, gr):
        # (the loss gradient starts from y: the fused and un-fused forwards agree to rounding, bit for bit on the
        # exact-fp32 kernels)
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-8, n
        assert torch.equal(a, b) or split, n
    # an in-place parameter update invalidates the shared weight packing
    with torch.no_grad():
        for p in maf.parameters():
            p.add_(0.01 * torch.randn_like(p))
    y2, l2, g2 = step()
    monkeypatch.setattr(bw, '_SAVE_BYTES', 0)
    maf._dev.pop(('bwd_weights', str(x0.device)), None)
    y3, l3, g3 = step()
    for n, a, b in zip(names, g2, g3):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-8, n
    assert float((g2[1] - gs[1]).abs().max()) > 1e-3 * float(gs[1].abs().max())


@pytest.mark.parametrize('weight_norm', [True, False])
def test_weight_norm_backward_prefix_kernel_matches_the_masked_one(weight_norm):
    """``tfep_weight_norm_backward_prefix`` (row of v staged in LDS, mask not read, packed gradient read over the live
    prefix) against ``tfep_weight_norm_backward`` reading the mask: same gradients to the rounding of the float64 sums
    (they run in another order), exact zeros on masked entries and fully masked rows, for permuted rows and columns."""
    from tfep_amd import _lib
    torch.manual_seed(9)
    N, K, ldw = 53, 9001, 9024
    dev = 'cuda'
    v = torch.randn(N, K, device=dev)
    wg = torch.rand(N, 1, device=dev) + 0.5 if weight_norm else None
    gW = torch.randn(N + 2, ldw, device=dev)
    in_of_col = torch.randperm(K, device=dev).to(torch.int32)
    col_of_in = torch.empty(K, dtype=torch.int32, device=dev)
    col_of_in[in_of_col.long()] = torch.arange(K, dtype=torch.int32, device=dev)
    row_of_out = torch.randperm(N + 2, device=dev)[:N].to(torch.int32)
    cut = torch.randint(0, K + 1, (N,), device=dev).to(torch.int32)
    cut[0], cut[1], cut[2] = 0, K, 5
    mask = (col_of_in[None, :].long() < cut[:, None].long()).float()
    lib = _lib.load()
    gv0, gg0 = torch.empty(N, K, device=dev), torch.empty(N, 1, device=dev)
    gv1, gg1 = torch.full((N, K), 7.0, device=dev), torch.full((N, 1), 7.0, device=dev)
    assert lib.tfep_weight_norm_backward(_lib.ptr(gW), ldw, _lib.ptr(v), _lib.ptr(wg), _lib.ptr(mask), N, K, _lib.ptr(row_of_out),
                                         _lib.ptr(col_of_in), _lib.ptr(gv0), _lib.ptr(gg0) if weight_norm else None, None) == 0
    assert lib.tfep_weight_norm_backward_prefix(_lib.ptr(gW), ldw, _lib.ptr(v), _lib.ptr(wg), N, K, _lib.ptr(row_of_out),
                                                _lib.ptr(in_of_col), _lib.ptr(cut), _lib.ptr(gv1),
                                                _lib.ptr(gg1) if weight_norm else None, None) == 0
    torch.cuda.synchronize()
    scale = float(gv0.abs().max())
    assert float((gv1 - gv0).abs().max()) <= 2e-6 * scale
    assert bool((gv1[mask == 0] == 0).all()) and bool((gv1[0] == 0).all())
    if weight_norm:
        assert torch.allclose(gg1, gg0, rtol=1e-6, atol=1e-7 * float(gg0.abs().max())) and float(gg1[0]) == 0.0
    # misuse
    assert lib.tfep_weight_norm_backward_prefix(_lib.ptr(gW), ldw, _lib.ptr(v), _lib.ptr(wg), N, K, None, _lib.ptr(in_of_col), None,
                                                _lib.ptr(gv1), _lib.ptr(gg1), None) != 0
    assert lib.tfep_weight_norm_backward_prefix(_lib.ptr(gW), K - 1, _lib.ptr(v), _lib.ptr(wg), N, K, None, _lib.ptr(in_of_col),
                                                _lib.ptr(cut), _lib.ptr(gv1), _lib.ptr(gg1), None) != 0


def test_training_steps_release_their_activations_without_the_garbage_collector():
    """The activations a training forward keeps must die with the graph, by reference counting: once they sat in a reference
    cycle (loss output -> grad_fn -> ctx.loss -> output; attributes of ctx) and 6.6 GB per cfg2 step waited for Python's
    cyclic collector.  With the collector off, device memory after several steps equals device memory after one."""
    import gc
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, _backward as bw
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(2)
    D, B = 64, 4096
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -4.0), torch.full((D,), 4.0), 8),
              hidden_layers=[512, 512], initialize_identity=False).cuda()
    x = torch.randn(B, D, device='cuda').clamp_(-3.9, 3.9)
    assert bw.saves_activations(maf, x)
    kept = B * 4 * (25 * D + 2 * 512)                          # bytes of parameters + hidden activations per step

    def step():
        for p in maf.parameters():
            p.grad = None
        y, l = maf(x)
        BoltzmannKLDivLoss()((y ** 2).sum(dim=1), l).backward()
    gc.collect()
    was_enabled = gc.isenabled()
    gc.disable()
    try:
        step()
        torch.cuda.synchronize()
        after_one = torch.cuda.memory_allocated()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        after_six = torch.cuda.memory_allocated()
    finally:
        if was_enabled:
            gc.enable()
    assert after_six - after_one < 0.5 * kept, (after_one, after_six, kept)


def test_cfg2_width_gradients_by_directional_derivative():
    """One cfg2 layer at full width (3000 features, 15 000 hidden units, 75 000 spline parameters) through the training
    path proper -- activations kept, split-f16 GEMMs, LDS-staged prefix packs and weight-norm backward, transposed operands
    as split rows: the gradient of the TFEP loss along a random direction in parameter space against the central
    difference of the loss itself, and the input gradient along a random direction in x."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF, _backward as bw
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(3)
    D, B = 3000, 384
    with torch.device('cuda'):
        maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), -5.0), torch.full((D,), 5.0), 8),
                  initialize_identity=False)
    assert maf._use_split_gemm()
    x = torch.randn(B, D, device='cuda').clamp_(-4.5, 4.5).requires_grad_(True)
    c = torch.rand(D, device='cuda') * 0.3
    loss_mod = BoltzmannKLDivLoss()

    def loss_of(xx):
        y, l = maf(xx)
        return loss_mod((c * y ** 2).sum(dim=1), l)
    assert bw.saves_activations(maf, x)
    loss = loss_of(x)
    loss.backward()
    params = [p for p in maf.parameters() if p.requires_grad]
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in params)
    gen = torch.Generator(device='cuda').manual_seed(8)
    # input gradient: along a direction in the first 300 features (they feed every later output through the conditioner);
    # over all 3000 features at once the loss of a random-init layer this wide is too rough for a central difference (its
    # value moves 10 % between steps of 1e-2 and 1e-3), which says nothing about the gradient
    dx = torch.zeros(B, D, device='cuda')
    dx[:, :300] = torch.randn(B, 300, device='cuda', generator=gen) * 3.0
    analytic_x = float((x.grad.double() * dx.double()).sum())

    def loss64(xx):                       # the same loss (mean of u_B - log|det J|) accumulated in float64: a float32 loss of
        y, l = maf(xx)                    # ~450 resolves 3e-5 of itself, which is the size of the differences taken here
        return float(((c * y ** 2).sum(dim=1).double() - l.double()).mean())
    assert abs(loss64(x.detach()) - float(loss)) < 1e-3 * abs(float(loss))
    # one direction per parameter tensor (a sum over all nine cancels to a tenth of its terms: their 0.5 % errors would read
    # as 4 %), errors measured against the largest of the nine derivatives
    got = []
    for p in params:
        d = torch.randn(p.shape, device='cuda', generator=gen) * p.detach().abs().mean()
        analytic = float((p.grad.double() * d.double()).sum())
        eps = 0.005
        with torch.no_grad():
            p.add_(d, alpha=eps)
            lp = loss64(x.detach())
            p.add_(d, alpha=-2 * eps)
            lm = loss64(x.detach())
            p.add_(d, alpha=eps)
        got.append((analytic, (lp - lm) / (2 * eps)))
    scale = max(abs(a) for a, _ in got)
    for (a, fd), (name, _) in zip(got, maf.named_parameters()):
        assert abs(fd - a) <= 0.015 * scale, (name, a, fd)
    with torch.no_grad():
        eps_x = 3e-3
        lp = loss64((x.detach() + eps_x * dx).clamp(-4.99, 4.99))
        lm = loss64((x.detach() - eps_x * dx).clamp(-4.99, 4.99))
    fd_x = (lp - lm) / (2 * eps_x)
    assert abs(fd_x - analytic_x) <= 0.03 * abs(analytic_x) + 1e-3, (fd_x, analytic_x)


@pytest.mark.gpu
@pytest.mark.parametrize('embedding', [False, True])
def test_parameter_gradients_do_not_depend_on_whether_the_input_needs_one(embedding):
    """The first layer of a flow sees data that needs no gradient: its backward skips the grad_input GEMM of the first masked
    linear (and the embedding's VJP).  The parameter gradients are the same bits either way."""
    from tfep_amd.loss import BoltzmannKLDivLoss
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.embeddings import PeriodicEmbedding
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import NeuralSplineTransformer
    torch.manual_seed(8)
    D, B = 60, 700
    emb = PeriodicEmbedding(D, limits=[0.0, 2.0], periodic_indices=list(range(0, D, 3))) if embedding else None
    lo, hi = (0.0, 2.0) if embedding else (-4.0, 4.0)
    maf = MAF(generate_degrees(D, 'ascending'), transformer=NeuralSplineTransformer(torch.full((D,), lo), torch.full((D,), hi), 8,
                                                                                   circular=embedding),
              embedding=emb, hidden_layers=[140, 120], initialize_identity=False).cuda()
    x0 = (torch.rand(B, D, device='cuda') * (hi - lo) + lo) * 0.97
    c = torch.linspace(0.1, 0.4, D, device='cuda')
    for split in (False, True):
        maf.split_gemm = split
        out = {}
        for needs in (True, False):
            for p in maf.parameters():
                p.grad = None
            x = x0.clone().requires_grad_(needs)
            y, l = maf(x)
            BoltzmannKLDivLoss()((c * y ** 2).sum(dim=1), l).backward()
            out[needs] = [p.grad.clone() for p in maf.parameters()]
            assert (x.grad is not None) == needs
        for a, b in zip(out[True], out[False]):
            assert torch.equal(a, b)


def test_inverse_of_a_mixed_transformer_with_a_moebius_member_is_differentiable():
    """Reference mixed.py:165-186 + moebius.py:142-147: autograd goes through every member of a MixedTransformer, and the
    Moebius inverse is the forward map on negated parameters.  Here: member by member on the HIP kernels
    (``_backward._differentiable_inverse``).  Checked by central differences along random directions (float32 kernels: a
    smooth loss, eps = 2e-3, 2 % tolerance) for the input and for every parameter at once, and against the same layer
    with the members in two separate layers' worth of plain autograd pieces (the values of the inverse itself)."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import MixedTransformer, MoebiusTransformer, NeuralSplineTransformer
    D, B = 10, 64
    torch.manual_seed(2)
    moeb_idx, spl_idx = [0, 1, 4, 5], [2, 3, 6, 7, 8, 9]
    deg = generate_degrees(D, 'ascending', repeats=2)
    tr = MixedTransformer([MoebiusTransformer(dimension=2, unit_sphere=False),
                           NeuralSplineTransformer(torch.full((6,), -3.0), torch.full((6,), 3.0), 4)], [moeb_idx, spl_idx])
    layer = MAF(deg, transformer=tr, hidden_layers=[48, 48], initialize_identity=False).cuda()
    y = (torch.randn(B, D, generator=torch.Generator().manual_seed(9)) * 0.8).cuda()
    c = torch.linspace(0.2, 1.0, D, device='cuda')
    params = [p for p in layer.parameters() if p.requires_grad]

    def loss(y_):
        x, ldj = layer.inverse(y_)
        return ((c * x ** 2).sum(dim=1) + ldj).mean()

    y_req = y.clone().requires_grad_(True)
    val = loss(y_req)
    val.backward()                                          # (raised NotImplementedError before round 4)
    gy = y_req.grad.clone()
    gp = [p.grad.clone() for p in params]
    assert all(bool(torch.isfinite(g).all()) for g in [gy] + gp) and float(gy.abs().max()) > 0
    with torch.no_grad():
        assert abs(float(val) - float(loss(y))) < 1e-6 * max(1.0, abs(float(val)))     # same values with and without a graph
        gen = torch.Generator(device='cuda').manual_seed(4)
        eps = 2e-3
        vy = torch.randn(B, D, device='cuda', generator=gen)
        fd = (float(loss(y + eps * vy)) - float(loss(y - eps * vy))) / (2 * eps)
        an = float((gy * vy).sum())
        assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-3), (fd, an)
        vp = [torch.randn(p.shape, device='cuda', generator=gen) * (p.abs().mean() + 1e-3) for p in params]
        an = float(sum((g * v).sum() for g, v in zip(gp, vp)))
        for s_ in (+1, -1):
            for p, v in zip(params, vp):
                p.add_(s_ * eps * v)
            layer._conditioner.invalidate_plan() if hasattr(layer._conditioner, 'invalidate_plan') else None
            if s_ > 0:
                lp = float(loss(y))
                for p, v in zip(params, vp):
                    p.sub_(eps * v)
            else:
                lm = float(loss(y))
                for p, v in zip(params, vp):
                    p.add_(eps * v)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - an) <= 2e-2 * max(abs(an), 1e-3), (fd, an)
