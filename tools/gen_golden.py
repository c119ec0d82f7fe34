#!/usr/bin/env python
"""Generate the golden vectors under ``tests/golden/`` from the reference itself.

DEV CONTAINER ONLY.  Imports the read-only reference (``/root/reference``) through
``tools/ref_shim.py`` and dumps inputs + expected outputs as small ``.npz`` /
``.json`` fixtures.  Only DATA is written (inputs, weights, outputs); no reference
source travels.  Run:  ``PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py``

For every floating-point case two reference outputs are stored:
  * ``*_f64``: the reference run in float64 on the float32-rounded inputs/weights
    (the parity target),
  * ``*_f32``: the reference run in float32 (its own noise floor vs float64).
"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_shim  # noqa: F401,E402

from tfep.nn import masked as rmasked  # noqa: E402
from tfep.nn.conditioners.made import MADE, generate_degrees  # noqa: E402
from tfep.nn.embeddings.mafembed import FlipInvariantEmbedding, MixedEmbedding, PeriodicEmbedding  # noqa: E402
from tfep.nn.flows.maf import MAF  # noqa: E402
from tfep.nn.flows.sequential import SequentialFlow  # noqa: E402
from tfep.nn.transformers.affine import AffineTransformer, VolumePreservingShiftTransformer  # noqa: E402
from tfep.nn.transformers.mixed import MixedTransformer  # noqa: E402
from tfep.nn.transformers.moebius import MoebiusTransformer  # noqa: E402
from tfep.nn.transformers.spline import NeuralSplineTransformer  # noqa: E402
from tfep.loss import BoltzmannKLDivLoss  # noqa: E402
from tfep.analysis.estimator import fep_estimator  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
os.makedirs(OUT, exist_ok=True)


def npy(t):
    return t.detach().cpu().numpy()


def gen(seed):
    g = torch.Generator()
    g.manual_seed(seed)
    return g


class f64:
    """Context: torch default dtype = float64."""
    def __enter__(self):
        self.old = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)

    def __exit__(self, *a):
        torch.set_default_dtype(self.old)


def sd_to_np(sd, prefix='sd/'):
    return {prefix + k: npy(v) for k, v in sd.items()}


def to_double_sd(sd):
    return {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


# -----------------------------------------------------------------------------
# 1. integer tables: degrees, hidden degrees, masks
# -----------------------------------------------------------------------------

def gen_degrees():
    cases = []
    # The argument sets below are the known-answer inputs of the reference's own
    # test (tests/nn/conditioners/test_made.py:50-67) plus docstring examples and
    # north-star sizes; expected values are produced by running the reference here.
    arglist = [
        dict(n_features=3),
        dict(n_features=7, order='descending'),
        dict(n_features=7, order='descending', max_value=2),
        dict(n_features=7, max_value=2, conditioning_indices=[0, 2, 3]),
        dict(n_features=6, repeats=2),
        dict(n_features=7, repeats=[1, 3, 2], conditioning_indices=[2]),
        dict(n_features=5, order='ascending', conditioning_indices=[0, 1]),
        dict(n_features=5, order='descending', conditioning_indices=[0, 3]),
        dict(n_features=7, order='ascending', max_value=3),
        dict(n_features=8, order='descending', repeats=2),
        dict(n_features=8, order='ascending', repeats=3),
        dict(n_features=9, order='descending', repeats=[2, 3, 1], max_value=2),
        dict(n_features=10, order='ascending', repeats=2, conditioning_indices=[1, 4, 7]),
        dict(n_features=66, order='ascending'),
        dict(n_features=66, order='descending'),
        dict(n_features=1024, order='ascending', repeats=2),
    ]
    for kw in arglist:
        cases.append({'kwargs': kw, 'expected': generate_degrees(**kw).tolist()})

    hidden_cases = []
    hidden_args = [
        # (degrees_in, degrees_out, hidden_layers)
        ([0, 1, 2], [0, 1, 2] * 2, 2),
        ([0, 1, 2, 3], [0, 0, 1, 1, 2, 2, 3, 3], 1),
        ([-1, -1, 0, 1, 2], [0, 1, 2, 0, 1, 2], 3),
        ([2, 1, 0], [2, 1, 0, 2, 1, 0], [4, 5]),
        ([-1, 0, 0, 1, 1], [0, 0, 1, 1], [6]),
        ([0, 1, 2, 3, 4], [0, 1, 2, 3, 4] * 3, [[0, 1, 2, 3], [3, 2, 1, 0, 0]]),
        ([-1, -1], [0, 0, 0, 0], 3),
        (generate_degrees(66).tolist(), generate_degrees(66).tile((2,)).tolist(), 2),
        (generate_degrees(12, order='descending').tolist(),
         generate_degrees(12, order='descending').tile((25,)).tolist(), 2),
    ]
    for din, dout, hl in hidden_args:
        out = MADE._get_degrees_hidden(torch.tensor(din), torch.tensor(dout), hl)
        hidden_cases.append({'degrees_in': din, 'degrees_out': dout, 'hidden_layers': hl,
                             'expected': [o.tolist() for o in out]})

    mask_cases = []
    for din, dout, strict, tr in [
        ([0, 1, 2], [0, 1, 2, 2], True, False),
        ([0, 1, 2], [0, 1, 2, 2], False, False),
        ([-1, 0, 1, 1], [0, 1, 2], True, True),
        ([-1, 0, 1, 1], [1, 0, 1, 0, -1], False, True),
        ([2, 1, 0], [2, 1, 0, 2, 1, 0], True, True),
    ]:
        m = rmasked.create_autoregressive_mask(np.array(din), np.array(dout), strictly_less=strict,
                                               transpose=tr, dtype=torch.float32)
        mask_cases.append({'degrees_in': din, 'degrees_out': dout, 'strictly_less': strict,
                           'transpose': tr, 'expected': m.int().tolist()})

    # Full MADE masks for a small net: bit-packed.
    made = MADE(degrees_in=generate_degrees(7, conditioning_indices=[1]),
                degrees_out=generate_degrees(6).tile((3,)), hidden_layers=2)
    made_masks = [npy(l.mask).astype(int).tolist() for l in made.layers[::2]]
    # Mask nnz at north-star-like structure but reduced size.
    nnz = []
    for D, P, order in [(66, 2, 'ascending'), (66, 2, 'descending'), (300, 25, 'ascending')]:
        deg = generate_degrees(D, order=order)
        m = MADE(degrees_in=deg, degrees_out=deg.tile((P,)), hidden_layers=2, weight_norm=False)
        nnz.append({'D': D, 'P': P, 'order': order,
                    'shapes': [list(l.mask.shape) for l in m.layers[::2]],
                    'nnz': [int(l.mask.sum().item()) for l in m.layers[::2]]})

    with open(os.path.join(OUT, 'degrees.json'), 'w') as f:
        json.dump({'generate_degrees': cases, 'degrees_hidden': hidden_cases,
                   'masks': mask_cases,
                   'made_masks': {'degrees_in': generate_degrees(7, conditioning_indices=[1]).tolist(),
                                  'degrees_out': generate_degrees(6).tile((3,)).tolist(),
                                  'hidden_layers': 2, 'expected': made_masks},
                   'mask_nnz': nnz}, f)


# -----------------------------------------------------------------------------
# 2. masked linear, weight norm, MADE
# -----------------------------------------------------------------------------

def gen_masked():
    out = {}
    g = gen(7)
    B, I, O = 6, 8, 5
    x = torch.randn(B, I, generator=g)
    w = torch.randn(O, I, generator=g)
    b = torch.randn(O, generator=g)
    mask = torch.tril(torch.ones(O, I))
    mask[2] = 0.0                                   # fully-masked row (NaN-safe path)
    out.update(x=npy(x), weight=npy(w), bias=npy(b), mask=npy(mask))
    out['y_f32'] = npy(rmasked.masked_linear(x, w, b, mask))
    out['y_f64'] = npy(rmasked.masked_linear(x.double(), w.double(), b.double(), mask.double()))
    out['y_nomask_f64'] = npy(rmasked.masked_linear(x.double(), w.double(), b.double(), None))

    # Weight norm module with the fully-masked row.
    torch.manual_seed(3)
    lin = rmasked.MaskedLinear(I, O, bias=True, mask=mask.clone())
    lin = rmasked.masked_weight_norm(lin, name='weight')
    with torch.no_grad():
        lin.weight_g.mul_(1.7)                      # make g != ||v||
    y32 = lin(x)
    out['wn_g'] = npy(lin.weight_g)
    out['wn_v'] = npy(lin.weight_v)
    out['wn_bias'] = npy(lin.bias)
    out['wn_weight_f32'] = npy(lin.weight)
    out['wn_y_f32'] = npy(y32)
    with f64():
        lin64 = rmasked.MaskedLinear(I, O, bias=True, mask=mask.double())
        lin64 = rmasked.masked_weight_norm(lin64, name='weight')
        lin64.load_state_dict(to_double_sd(lin.state_dict()))
        y64 = lin64(x.double())
        out['wn_weight_f64'] = npy(lin64.weight)
        out['wn_y_f64'] = npy(y64)
    out['wn_state_keys'] = np.array(sorted(lin.state_dict().keys()))
    np.savez_compressed(os.path.join(OUT, 'masked_linear.npz'), **out)


def run_made(kwargs, x, seed):
    torch.manual_seed(seed)
    m32 = MADE(**kwargs)
    with torch.no_grad():
        for l in m32.layers[::2]:               # decouple g from ||v|| and bias from init
            if hasattr(l, 'weight_g'):
                l.weight_g.mul_(torch.rand_like(l.weight_g) + 0.5)
    y32 = m32(x)
    with f64():
        m64 = MADE(**kwargs)
        m64.load_state_dict(to_double_sd(m32.state_dict()))
        h = x.double()
        hidden = []
        for mod in m64.layers:
            h = mod(h)
            if isinstance(mod, torch.nn.ELU):
                hidden.append(npy(h))
        y64 = h
    return m32, npy(y32), npy(y64), hidden


def gen_made():
    cases = {
        'a': dict(degrees_in=generate_degrees(5), degrees_out=generate_degrees(5).tile((2,)),
                  hidden_layers=2, weight_norm=True),
        'b': dict(degrees_in=generate_degrees(7, conditioning_indices=[0, 3]),
                  degrees_out=generate_degrees(5).tile((3,)), hidden_layers=1, weight_norm=False),
        'c': dict(degrees_in=generate_degrees(6, order='descending'),
                  degrees_out=generate_degrees(6, order='descending').tile((4,)),
                  hidden_layers=[7, 9], weight_norm=True),
        'd': dict(degrees_in=generate_degrees(8, repeats=2),
                  degrees_out=generate_degrees(8, repeats=2).tile((2,)),
                  hidden_layers=3, weight_norm=True),
    }
    out = {}
    meta = {}
    for name, kw in cases.items():
        n_in = len(kw['degrees_in'])
        x = torch.randn(9, n_in, generator=gen(11))
        m32, y32, y64, hidden = run_made(kw, x, seed=5)
        out.update(sd_to_np(m32.state_dict(), prefix=f'{name}/sd/'))
        out[f'{name}/x'] = npy(x)
        out[f'{name}/y_f32'] = y32
        out[f'{name}/y_f64'] = y64
        for i, h in enumerate(hidden):
            out[f'{name}/hidden{i}_f64'] = h
        meta[name] = dict(degrees_in=kw['degrees_in'].tolist(), degrees_out=kw['degrees_out'].tolist(),
                          hidden_layers=kw['hidden_layers'], weight_norm=kw['weight_norm'],
                          n_parameters=int(m32.n_parameters()))
    out['meta'] = np.array(json.dumps(meta))
    np.savez_compressed(os.path.join(OUT, 'made.npz'), **out)


# -----------------------------------------------------------------------------
# 3. transformers
# -----------------------------------------------------------------------------

SPLINE_VARIANTS = [
    # (circular, identity_boundary_slopes, learn_lower, learn_upper)
    (False, False, False, False),
    (False, True, False, False),
    (False, False, True, False),
    (False, False, False, True),
    (False, False, True, True),
    (False, True, True, False),
    (False, True, False, True),
    (False, True, True, True),
    (True, False, False, False),
    (True, True, False, False),
]


def variant_name(v):
    return 'c%d_i%d_l%d_u%d' % tuple(int(b) for b in v)


def run_transformer(make, x, par):
    """Run forward + inverse in f32 and f64; return dict of arrays."""
    out = {}
    t32 = make(torch.float32)
    y32, l32 = t32(x, par)
    out['y_f32'], out['ldj_f32'] = npy(y32), npy(l32)
    with f64():
        t64 = make(torch.float64)
        y64, l64 = t64(x.double(), par.double())
        # The inverse is evaluated on a float32-representable input (the rounded forward output):
        # where the map is flat, x = f^-1(y) amplifies the rounding of y by 1/f', so a float32
        # implementation can only be compared on inputs it can represent.
        yin = y64.detach().float()
        xi, li = t64.inverse(yin.double(), par.double())
        out['y_f64'], out['ldj_f64'] = npy(y64), npy(l64)
        out['inv_in'] = npy(yin)
        out['xinv_f64'], out['ldjinv_f64'] = npy(xi), npy(li)
    return out


def gen_transformers():
    out = {}
    B, D = 16, 6

    # ---- affine
    g = gen(21)
    x = torch.randn(B, D, generator=g)
    par = torch.randn(B, 2 * D, generator=g) * 0.7
    res = run_transformer(lambda dt: AffineTransformer(), x, par)
    out.update({'affine/x': npy(x), 'affine/par': npy(par)})
    out.update({'affine/' + k: v for k, v in res.items()})

    # ---- volume preserving shift (with periodic wrap)
    g = gen(22)
    x = torch.rand(B, D, generator=g) * 2 - 1
    par = torch.randn(B, D, generator=g)
    pidx = torch.tensor([1, 4])
    plim = torch.tensor([-1.0, 1.0])
    res = run_transformer(lambda dt: VolumePreservingShiftTransformer(
        periodic_indices=pidx, periodic_limits=plim.to(dt)), x, par)
    out.update({'volpres/x': npy(x), 'volpres/par': npy(par),
                'volpres/periodic_indices': npy(pidx), 'volpres/periodic_limits': npy(plim)})
    out.update({'volpres/' + k: v for k, v in res.items()})

    # ---- splines: all legal flag combinations, K = 8 and a K = 3 case with y-domain != x-domain
    spl_meta = {}
    for K, dom in [(8, 'sym'), (3, 'asym')]:
        for v in SPLINE_VARIANTS:
            circ, ident, ll, lu = v
            name = f'spline/K{K}_{dom}_{variant_name(v)}'
            if dom == 'sym':
                x0 = torch.full((D,), -5.0)
                xf = torch.full((D,), 5.0)
                y0 = yf = None
            else:
                x0 = torch.tensor([-2.0, -1.0, 0.0, -3.0, -0.5, 1.0])
                xf = torch.tensor([0.0, 1.5, 2.0, 3.0, 0.5, 4.0])
                if circ:
                    y0 = yf = None
                else:
                    y0 = x0 + 1.0
                    yf = y0 + (xf - x0) * 1.5
            kw = dict(n_bins=K, circular=circ, identity_boundary_slopes=ident,
                      learn_lower_bound=ll, learn_upper_bound=lu)

            def make(dt, x0=x0, xf=xf, y0=y0, yf=yf, kw=kw):
                return NeuralSplineTransformer(
                    x0=x0.to(dt), xf=xf.to(dt),
                    y0=None if y0 is None else y0.to(dt),
                    yf=None if yf is None else yf.to(dt), **kw)

            t = make(torch.float32)
            P = int(t.n_parameters_per_feature)
            g = gen(100 + K + 7 * SPLINE_VARIANTS.index(v))
            par = torch.randn(B, P * D, generator=g)
            if ll or lu:
                # Keep the learnable-domain scale/shift moderate.
                pv = par.reshape(B, P, D)
                pv[:, -1] *= 0.2
                if ll and lu:
                    pv[:, -2] *= 0.5
                par = pv.reshape(B, P * D).contiguous()
            # Inputs: rows 0..B-5 in-domain randoms, then out-of-domain on both sides,
            # then exactly-on-knot values (filled below).
            width = xf - x0
            if circ:
                x = torch.rand(B, D, generator=g) * width + x0
            else:
                x = (torch.rand(B, D, generator=g) * 0.96 + 0.02) * width + x0
                x[-4] = x0 - 0.37 * width          # below the domain
                x[-3] = xf + 0.21 * width          # above the domain
            # Exactly-on-knot inputs (strict '>' puts them in the lower bin): use the f32 knots.
            nx0, ny0, w_, h_, s_, sh_ = t._get_parameters(par)
            base = nx0 if nx0.dim() == 1 else nx0.unsqueeze(1)
            knots = base + torch.cumsum(w_, dim=1)          # (B, K, D) upper knots of each bin
            if not circ:
                x[-2] = knots[-2, 1]                         # an interior knot
                x[-1] = (nx0 if nx0.dim() == 1 else nx0[-1])  # the first knot x0
            res = run_transformer(make, x, par)
            with f64():
                t64 = make(torch.float64)
                nx0, ny0, w_, h_, s_, sh_ = t64._get_parameters(par.double())
                res['gp_x0_f64'] = npy(nx0 if nx0.dim() == 2 else nx0.expand(B, D))
                res['gp_y0_f64'] = npy(ny0 if ny0.dim() == 2 else ny0.expand(B, D))
                res['gp_widths_f64'], res['gp_heights_f64'], res['gp_slopes_f64'] = npy(w_), npy(h_), npy(s_)
                if sh_ is not None:
                    res['gp_shifts_f64'] = npy(sh_)
            out.update({name + '/x': npy(x), name + '/par': npy(par)})
            out.update({name + '/' + k: vv for k, vv in res.items()})
            spl_meta[name] = dict(x0=x0.tolist(), xf=xf.tolist(),
                                  y0=None if y0 is None else y0.tolist(),
                                  yf=None if yf is None else yf.tolist(),
                                  P=P, **kw)
    out['spline/meta'] = np.array(json.dumps(spl_meta))

    # ---- Moebius, d = 2 and 3, unit_sphere both
    mo_meta = {}
    for d in (2, 3):
        for unit in (False, True):
            name = f'moebius/d{d}_u{int(unit)}'
            g = gen(300 + d + int(unit))
            nv = 4
            x = torch.randn(B, nv * d, generator=g)
            if unit:
                xv = x.reshape(B, nv, d)
                x = (xv / torch.linalg.norm(xv, dim=-1, keepdim=True)).reshape(B, nv * d)
                x = x.double().float()
            par = torch.randn(B, nv * d, generator=g)
            res = run_transformer(lambda dt: MoebiusTransformer(dimension=d, unit_sphere=unit), x, par)
            out.update({name + '/x': npy(x), name + '/par': npy(par)})
            out.update({name + '/' + k: vv for k, vv in res.items()})
            mo_meta[name] = dict(dimension=d, unit_sphere=unit, max_radius=0.99)
    out['moebius/meta'] = np.array(json.dumps(mo_meta))

    # ---- Mixed: spline on [0,2,5], affine on [1,3], moebius(d=2, unit) on... keep simple 2+: affine+spline+volpres
    g = gen(41)
    Dm = 7
    ind = [[0, 2, 5], [1, 3], [4, 6]]
    x = torch.rand(B, Dm, generator=g) * 1.6 - 0.8

    def make_mixed(dt):
        return MixedTransformer(
            transformers=[
                NeuralSplineTransformer(x0=torch.full((3,), -1.0).to(dt), xf=torch.full((3,), 1.0).to(dt), n_bins=4),
                AffineTransformer(),
                NeuralSplineTransformer(x0=torch.full((2,), -1.0).to(dt), xf=torch.full((2,), 1.0).to(dt),
                                        n_bins=3, circular=True),
            ], indices=ind)
    tm = make_mixed(torch.float32)
    npar = len(tm.get_identity_parameters(Dm))
    par = torch.randn(B, npar, generator=g)
    res = run_transformer(make_mixed, x, par)
    out.update({'mixed/x': npy(x), 'mixed/par': npy(par)})
    out.update({'mixed/' + k: v for k, v in res.items()})
    out['mixed/degrees_out'] = npy(tm.get_degrees_out(torch.arange(Dm)))
    out['mixed/split'] = npy(tm._parameters_split_indices)

    # ---- periodic embedding
    g = gen(51)
    x = torch.rand(B, 6, generator=g)
    emb = PeriodicEmbedding(n_features_in=6, limits=[0.0, 1.0], periodic_indices=[1, 2, 5])
    out['pemb/x'] = npy(x)
    out['pemb/y_f32'] = npy(emb(x))
    with f64():
        emb64 = PeriodicEmbedding(n_features_in=6, limits=[0.0, 1.0], periodic_indices=[1, 2, 5])
        out['pemb/y_f64'] = npy(emb64(x.double()))
    out['pemb/degrees_in'] = np.array([0, 1, 2, 3, 4, 5])
    out['pemb/degrees_out'] = npy(emb.get_degrees_out(torch.arange(6)))
    out['pemb/periodic_indices'] = npy(emb._periodic_indices)
    out['pemb/nonperiodic_indices'] = npy(emb._nonperiodic_indices)

    np.savez_compressed(os.path.join(OUT, 'transformers.npz'), **out)


# -----------------------------------------------------------------------------
# 4. flows (end to end)
# -----------------------------------------------------------------------------

def perturb_weight_g(flow, seed):
    g = gen(seed)
    with torch.no_grad():
        for n, p in flow.named_parameters():
            if n.endswith('weight_g'):
                p.mul_(torch.rand(p.shape, generator=g) + 0.5)


def run_flow(make_flow, x, seed, out, name, inverse=True, store_masks=False):
    torch.manual_seed(seed)
    f32 = make_flow(torch.float32)
    perturb_weight_g(f32, seed + 1)
    # Taken before the first pass: the reference's PartialFlow registers a new buffer lazily in forward().
    sd0 = {k: v.clone() for k, v in f32.state_dict().items()}
    y32, l32 = f32(x)
    with f64():
        f64m = make_flow(torch.float64)
        f64m.load_state_dict(to_double_sd(sd0))
        y64, l64 = f64m(x.double())
        out[f'{name}/y_f64'], out[f'{name}/ldj_f64'] = npy(y64), npy(l64)
        if inverse:
            # Inverse of the float32-rounded forward output, in float64.
            yin = y32.detach().double()
            xi, li = f64m.inverse(yin)
            out[f'{name}/inv_in'] = npy(y32)
            out[f'{name}/xinv_f64'], out[f'{name}/ldjinv_f64'] = npy(xi), npy(li)
    out[f'{name}/x'] = npy(x)
    out[f'{name}/y_f32'], out[f'{name}/ldj_f32'] = npy(y32), npy(l32)
    for k, v in sd0.items():
        if k.endswith('.mask') and not store_masks:
            continue
        if k.endswith('.mask'):
            out[f'{name}/sdmask/{k}'] = np.packbits(npy(v).astype(bool), axis=None)
            out[f'{name}/sdmaskshape/{k}'] = np.array(v.shape)
        else:
            out[f'{name}/sd/{k}'] = npy(v)
    out[f'{name}/n_parameters'] = np.array(int(f32.n_parameters()))
    return f32


def gen_flows():
    out = {}

    # cfg1 exactly: 2-layer MAF + affine, D=66, B=1024, weight_norm, no identity init.
    D = 66
    x = torch.randn(1024, D, generator=gen(1234))

    def make_cfg1(dt):
        return SequentialFlow(
            MAF(degrees_in=generate_degrees(D, order='ascending'), initialize_identity=False),
            MAF(degrees_in=generate_degrees(D, order='descending'), initialize_identity=False),
        )
    run_flow(make_cfg1, x, 0, out, 'cfg1', inverse=False, store_masks=True)
    # inverse on a small slice (D sequential passes)
    torch.manual_seed(0)

    # 4-layer RQ-8, D=66, hidden [96, 96]; in-domain and tail inputs.
    def make_rq(dt):
        layers = []
        for i in range(4):
            layers.append(MAF(
                degrees_in=generate_degrees(D, order='ascending' if i % 2 == 0 else 'descending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -5.0).to(dt),
                                                    xf=torch.full((D,), 5.0).to(dt), n_bins=8),
                hidden_layers=[96, 96], initialize_identity=False))
        return SequentialFlow(*layers)
    x_in = torch.randn(192, D, generator=gen(1234)).clamp(-4.9, 4.9)
    x_tail = 1.5 * torch.randn(64, D, generator=gen(4321)) * 2.0
    x = torch.cat([x_in, x_tail])
    run_flow(make_rq, x, 0, out, 'rq4', inverse=True)

    # Conditioning DOFs + repeats + default hidden width, affine; D=10.
    def make_cond(dt):
        deg = generate_degrees(10, conditioning_indices=[0, 7], repeats=2)
        return SequentialFlow(
            MAF(degrees_in=deg, initialize_identity=False),
            MAF(degrees_in=generate_degrees(10, order='descending', conditioning_indices=[0, 7], repeats=2),
                transformer=NeuralSplineTransformer(x0=torch.full((8,), -4.0).to(dt),
                                                    xf=torch.full((8,), 4.0).to(dt), n_bins=5,
                                                    identity_boundary_slopes=True),
                initialize_identity=False, weight_norm=False),
        )
    x = torch.randn(40, 10, generator=gen(99))
    run_flow(make_cond, x, 2, out, 'cond', inverse=True, store_masks=True)

    # Circular spline + periodic embedding on 8 angles in [0, 1) (cfg4-i recipe, reduced).
    def make_circ(dt):
        layers = []
        for i in range(2):
            layers.append(MAF(
                degrees_in=generate_degrees(8, order='ascending' if i % 2 == 0 else 'descending'),
                transformer=NeuralSplineTransformer(x0=torch.zeros(8).to(dt), xf=torch.ones(8).to(dt),
                                                    n_bins=8, circular=True),
                embedding=PeriodicEmbedding(n_features_in=8, limits=[0.0, 1.0]),
                initialize_identity=False))
        return SequentialFlow(*layers)
    x = torch.rand(48, 8, generator=gen(77))
    run_flow(make_circ, x, 4, out, 'circ', inverse=True)

    # Moebius d=2 unit sphere on 6 angles -> 12 features, repeats=2 (cfg4-ii recipe, reduced).
    def make_moeb(dt):
        layers = []
        for i in range(2):
            layers.append(MAF(
                degrees_in=generate_degrees(12, order='ascending' if i % 2 == 0 else 'descending', repeats=2),
                transformer=MoebiusTransformer(dimension=2, unit_sphere=True),
                initialize_identity=False))
        return SequentialFlow(*layers)
    ang = torch.rand(32, 6, generator=gen(78)) * 2 * np.pi
    x = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(32, 12)
    run_flow(make_moeb, x, 6, out, 'moeb', inverse=True)

    # Mixed transformer inside a MAF: spline on even, affine on odd features; D=6.
    def make_mixflow(dt):
        return SequentialFlow(MAF(
            degrees_in=generate_degrees(6),
            transformer=MixedTransformer(
                transformers=[
                    NeuralSplineTransformer(x0=torch.full((3,), -3.0).to(dt), xf=torch.full((3,), 3.0).to(dt), n_bins=4),
                    AffineTransformer()],
                indices=[[0, 2, 4], [1, 3, 5]]),
            initialize_identity=False))
    x = torch.randn(24, 6, generator=gen(79))
    run_flow(make_mixflow, x, 8, out, 'mixflow', inverse=True)

    # Identity initialisation: y == x, ldj == 0 (tests/nn/flows/test_maf.py:221-223).
    def make_ident(dt):
        return SequentialFlow(MAF(
            degrees_in=generate_degrees(5),
            transformer=NeuralSplineTransformer(x0=torch.full((5,), -2.0).to(dt), xf=torch.full((5,), 2.0).to(dt), n_bins=4),
            initialize_identity=True))
    x = torch.randn(8, 5, generator=gen(80))
    torch.manual_seed(10)
    fid = make_ident(torch.float32)
    y, l = fid(x)
    out['ident/x'] = npy(x)
    out['ident/y_f32'], out['ident/ldj_f32'] = npy(y), npy(l)
    for k, v in fid.state_dict().items():
        if not k.endswith('.mask'):
            out[f'ident/sd/{k}'] = npy(v)

    np.savez_compressed(os.path.join(OUT, 'flows.npz'), **out)


# -----------------------------------------------------------------------------
# 4b. gradients of one training step (reference autograd, float64)
# -----------------------------------------------------------------------------

def grad_flows(D=10):
    """The flow configurations of the gradient goldens (forward and inverse): name -> constructor(dtype)."""
    from tfep.nn.conditioners.made import generate_degrees as gd
    flows = {
        'affine': lambda dt: SequentialFlow(
            MAF(degrees_in=gd(D, 'ascending'), initialize_identity=False),
            MAF(degrees_in=gd(D, 'descending', conditioning_indices=[2, 5]), weight_norm=False,
                initialize_identity=False)),
        'spline': lambda dt: SequentialFlow(
            MAF(degrees_in=gd(D, 'ascending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -4.0).to(dt), xf=torch.full((D,), 4.0).to(dt), n_bins=8),
                initialize_identity=False),
            MAF(degrees_in=gd(D, 'descending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -4.0).to(dt), xf=torch.full((D,), 4.0).to(dt), n_bins=8),
                hidden_layers=[24, 24], initialize_identity=False)),
        'circular': lambda dt: SequentialFlow(
            MAF(degrees_in=gd(D, 'ascending'),
                transformer=NeuralSplineTransformer(x0=torch.zeros(D).to(dt), xf=torch.ones(D).to(dt), n_bins=8, circular=True),
                embedding=PeriodicEmbedding(n_features_in=D, limits=[0.0, 1.0]), initialize_identity=False)),
        'identslopes': lambda dt: SequentialFlow(
            MAF(degrees_in=gd(D, 'ascending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -2.0).to(dt), xf=torch.full((D,), 2.0).to(dt), n_bins=5,
                                                    identity_boundary_slopes=True),
                initialize_identity=False)),
    }
    flows['moebius'] = lambda dt: SequentialFlow(
        MAF(degrees_in=gd(12, 'ascending', repeats=2), transformer=MoebiusTransformer(dimension=2, unit_sphere=True),
            initialize_identity=False),
        MAF(degrees_in=gd(12, 'descending', repeats=3), transformer=MoebiusTransformer(dimension=3, unit_sphere=False),
            initialize_identity=False))
    flows['mixed'] = lambda dt: SequentialFlow(
        MAF(degrees_in=gd(D, 'ascending'),
            transformer=MixedTransformer(
                transformers=[
                    NeuralSplineTransformer(x0=torch.full((4,), -3.0).to(dt), xf=torch.full((4,), 3.0).to(dt), n_bins=4),
                    AffineTransformer(),
                    VolumePreservingShiftTransformer()],
                indices=[[0, 2, 4, 6], [1, 3, 5], [7, 8, 9]]),
            initialize_identity=False))
    for tag, (ll, lu) in {'learnlow': (True, False), 'learnup': (False, True), 'learnboth': (True, True)}.items():
        flows[tag] = (lambda ll, lu: lambda dt: SequentialFlow(
            MAF(degrees_in=gd(D, 'ascending'),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -3.0).to(dt), xf=torch.full((D,), 3.0).to(dt),
                                                    n_bins=6, learn_lower_bound=ll, learn_upper_bound=lu),
                initialize_identity=False)))(ll, lu)
    return flows


def gen_grads():
    """loss = BoltzmannKLDivLoss()(u_B(y), log_det_J) with the synthetic potential
    u_B(y) = sum_f (c_f y_f^2 + d_f y_f), backward through the reference flow in float64.
    Stores d loss / d(every trainable parameter) and d loss / d x."""
    from tfep.nn.conditioners.made import generate_degrees as gd
    out = {}
    D = 10

    def quad(y, c, d):
        return (c * y ** 2 + d * y).sum(dim=1)

    flows = grad_flows(D)
    for name, make in flows.items():
        torch.manual_seed(20)
        f32 = make(torch.float32)
        perturb_weight_g(f32, 21)
        B = 48
        g = gen(22)
        Dn = 12 if name == 'moebius' else D
        if name == 'circular':
            x = torch.rand(B, Dn, generator=g)
        elif name == 'moebius':
            ang = torch.rand(B, 6, generator=g) * 2 * np.pi
            x = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(B, 12)
        else:
            x = torch.randn(B, Dn, generator=g) * 1.5
            x[-1] = 6.0            # out-of-domain rows (both tails)
            x[-2] = -5.5
        c = torch.rand(Dn, generator=g) * 0.3
        d = torch.randn(Dn, generator=g) * 0.2
        with f64():
            m = make(torch.float64)
            m.load_state_dict(to_double_sd(f32.state_dict()))
            xd = x.double().requires_grad_(True)
            y, ldj = m(xd)
            loss = BoltzmannKLDivLoss()(quad(y, c.double(), d.double()), ldj)
            loss.backward()
            out[f'{name}/loss_f64'] = npy(loss)
            out[f'{name}/gx_f64'] = npy(xd.grad)
            out[f'{name}/y_f64'], out[f'{name}/ldj_f64'] = npy(y), npy(ldj)
            for k, p in m.named_parameters():
                out[f'{name}/grad/{k}'] = npy(p.grad)
        out[f'{name}/x'] = npy(x)
        out[f'{name}/c'], out[f'{name}/d'] = npy(c), npy(d)
        for k, v in f32.state_dict().items():
            if not k.endswith('.mask'):
                out[f'{name}/sd/{k}'] = npy(v)

    # masked linear alone (reference gradcheck target, tests/nn/test_masked.py:150-170)
    g = gen(30)
    x = torch.randn(7, 6, generator=g, dtype=torch.float64, requires_grad=True)
    w = torch.randn(5, 6, generator=g, dtype=torch.float64, requires_grad=True)
    b = torch.randn(5, generator=g, dtype=torch.float64, requires_grad=True)
    mask = torch.tril(torch.ones(5, 6, dtype=torch.float64))
    gy = torch.randn(7, 5, generator=g, dtype=torch.float64)
    yy = rmasked.masked_linear(x, w, b, mask)
    yy.backward(gy)
    out.update({'ml/x': npy(x), 'ml/w': npy(w), 'ml/b': npy(b), 'ml/mask': npy(mask), 'ml/gy': npy(gy),
                'ml/gx': npy(x.grad), 'ml/gw': npy(w.grad), 'ml/gb': npy(b.grad)})
    np.savez_compressed(os.path.join(OUT, 'grads.npz'), **out)


def gen_inv_grads():
    """Gradients THROUGH THE INVERSE (reference autoregressive.py:179-229 is plain differentiable torch): loss =
    BoltzmannKLDivLoss()(u(x), log_det_J_inv) with (x, log_det_J_inv) = flow.inverse(y) and u(x) = sum_f (c_f x_f^2 + d_f x_f),
    backward in float64.  Stores d loss / d y and d loss / d(every trainable parameter)."""
    out = {}
    D = 10

    def quad(y, c, d):
        return (c * y ** 2 + d * y).sum(dim=1)
    for name, make in grad_flows(D).items():
        torch.manual_seed(20)
        f32 = make(torch.float32)
        perturb_weight_g(f32, 21)
        B = 24
        g = gen(23)
        Dn = 12 if name == 'moebius' else D
        if name == 'circular':
            y = torch.rand(B, Dn, generator=g)
        elif name == 'moebius':
            ang = torch.rand(B, 6, generator=g) * 2 * np.pi
            y = torch.stack([torch.cos(ang), torch.sin(ang)], dim=2).reshape(B, 12)
        else:
            y = torch.randn(B, Dn, generator=g) * 1.5
            y[-1] = 5.0            # out-of-domain rows (both tails)
            y[-2] = -4.5
        c = torch.rand(Dn, generator=g) * 0.3
        d = torch.randn(Dn, generator=g) * 0.2
        with f64():
            m = make(torch.float64)
            m.load_state_dict(to_double_sd(f32.state_dict()))
            yd = y.double().requires_grad_(True)
            x, ldj = m.inverse(yd)
            loss = BoltzmannKLDivLoss()(quad(x, c.double(), d.double()), ldj)
            loss.backward()
            out[f'{name}/loss_f64'] = npy(loss)
            out[f'{name}/gy_f64'] = npy(yd.grad)
            out[f'{name}/x_f64'], out[f'{name}/ldj_f64'] = npy(x), npy(ldj)
            for k, p in m.named_parameters():
                out[f'{name}/grad/{k}'] = npy(p.grad)
        out[f'{name}/y'] = npy(y)
        out[f'{name}/c'], out[f'{name}/d'] = npy(c), npy(d)
        for k, v in f32.state_dict().items():
            if not k.endswith('.mask'):
                out[f'{name}/sd/{k}'] = npy(v)
    np.savez_compressed(os.path.join(OUT, 'inv_grads.npz'), **out)


def wide_parameters(module, seed):
    """Deterministic parameters for the wide gradient case, a function of (name order, shape, seed) only -- the test builds
    the same tensors on its side instead of shipping 2 x 14 M weights: v ~ U(-1, 1) / sqrt(fan_in) (masked by the module's
    own pre-hook), g ~ U(0.5, 1.5) x row norm of the masked v, bias ~ U(-1, 1) / sqrt(fan_in)."""
    with torch.no_grad():
        for i, (n, p) in enumerate(module.named_parameters()):
            g = gen(seed + i)
            if n.endswith('weight_v') or n.endswith('.weight'):
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) / p.shape[1] ** 0.5).to(p.dtype))
            elif n.endswith('weight_g'):
                p.copy_((torch.rand(p.shape, generator=g) + 0.5).to(p.dtype))         # (relative to norm 1 rows: see below)
            else:
                p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * 0.05).to(p.dtype))


def gen_grads_wide():
    """Gradients of a 2-layer RQ-8 MAF at D = 300 with the default hidden width (1498; 13.9 M weights per layer): the
    size at which the backward's k-ranges, prefix packs and split transposes are multi-tile.  The parameters are a
    function of a seed (``wide_parameters``), so only SAMPLES of the big gradients are stored: per tensor its float64 L2
    norm and maximum, 4096 random entries (index, float64 value, the reference's own float32 value) and the float32-vs-
    float64 relative L2 of the whole tensor (the noise floor); small tensors (biases, weight_g, d loss / d x) whole.
    Also one AdamW step from identical state: the parameter deltas at the same sampled entries."""
    out = {}
    D, B, K = 300, 256, 8

    def make(dt):
        return SequentialFlow(*[
            MAF(degrees_in=generate_degrees(D, order=o),
                transformer=NeuralSplineTransformer(x0=torch.full((D,), -5.0).to(dt), xf=torch.full((D,), 5.0).to(dt), n_bins=K),
                initialize_identity=False) for o in ('ascending', 'descending')])
    g = gen(41)
    x = (torch.randn(B, D, generator=g)).clamp_(-4.9, 4.9)
    c = torch.rand(D, generator=g) * 0.3
    d = torch.randn(D, generator=g) * 0.2

    def loss_of(m, xx, cc, dd):
        y, ldj = m(xx)
        return BoltzmannKLDivLoss()((cc * y ** 2 + dd * y).sum(dim=1), ldj)
    torch.manual_seed(40)
    f32 = make(torch.float32)
    wide_parameters(f32, 400)
    x32 = x.clone().requires_grad_(True)
    l32 = loss_of(f32, x32, c, d)
    l32.backward()
    with f64():
        m = make(torch.float64)
        m.load_state_dict(to_double_sd(f32.state_dict()))
        xd = x.double().requires_grad_(True)
        loss = loss_of(m, xd, c.double(), d.double())
        loss.backward()
        # one AdamW step (the optimiser of the reference's maps, app/base.py) from this state, float64
        opt = torch.optim.AdamW(m.parameters(), lr=1e-3, weight_decay=0.01)
        before = {k: p.detach().clone() for k, p in m.named_parameters()}
        opt.step()
    out['loss_f64'], out['loss_f32'] = npy(loss), npy(l32)
    out['x'], out['c'], out['d'] = npy(x), npy(c), npy(d)
    out['gx_f64'], out['gx_f32'] = npy(xd.grad), npy(x32.grad)
    p32 = dict(f32.named_parameters())
    for i, (k, p) in enumerate(m.named_parameters()):
        g64, g32 = p.grad, p32[k].grad.double()
        out[f'norm/{k}'] = npy(g64.norm())
        out[f'max/{k}'] = npy(g64.abs().max())
        out[f'noise/{k}'] = npy((g32 - g64).norm() / g64.norm())
        delta = (p.detach() - before[k])
        if p.numel() <= 16384:
            out[f'full/{k}'] = npy(g64)
            out[f'full32/{k}'] = npy(g32)
            out[f'adamw/{k}'] = npy(delta)
        else:
            idx = torch.randint(0, p.numel(), (4096,), generator=gen(500 + i))
            out[f'idx/{k}'] = npy(idx)
            out[f'val/{k}'] = npy(g64.flatten()[idx])
            out[f'val32/{k}'] = npy(g32.flatten()[idx])
            out[f'adamw/{k}'] = npy(delta.flatten()[idx])
    np.savez_compressed(os.path.join(OUT, 'grads_wide.npz'), **out)


# -----------------------------------------------------------------------------
# 5. loss + estimator
# -----------------------------------------------------------------------------

def gen_loss():
    out = {}
    g = gen(5)
    N = 257
    uB = torch.randn(N, generator=g) * 3 + 10
    ldj = torch.randn(N, generator=g)
    lw = torch.randn(N, generator=g)
    uA = torch.randn(N, generator=g) * 2 + 8
    out.update(uB=npy(uB), ldj=npy(ldj), lw=npy(lw), uA=npy(uA))
    import warnings
    warnings.simplefilter('ignore')
    for dt, tag in [(torch.float32, 'f32'), (torch.float64, 'f64')]:
        # The estimator takes log(N) in the torch DEFAULT dtype (estimator.py:75-77), so the
        # float64 goldens are produced under default dtype float64, like the reference's tests.
        torch.set_default_dtype(dt)
        a, b_, c, d = uB.to(dt), ldj.to(dt), lw.to(dt), uA.to(dt)
        L = BoltzmannKLDivLoss()
        out[f'loss_plain_{tag}'] = npy(L(a, b_))
        out[f'loss_ref_{tag}'] = npy(L(a, b_, ref_potentials=d))
        out[f'loss_weighted_{tag}'] = npy(L(a, b_, log_weights=c))
        out[f'loss_all_{tag}'] = npy(L(a, b_, log_weights=c, ref_potentials=d))
        out[f'loss_noldj_{tag}'] = npy(L(a))
        an = a.clone()
        an[[3, 77]] = float('nan')
        Ln = BoltzmannKLDivLoss(ignore_nan=True)
        out[f'loss_nan_plain_{tag}'] = npy(Ln(an, b_))
        out[f'loss_nan_weighted_{tag}'] = npy(Ln(an, b_, log_weights=c))
        out[f'loss_nan_propagates_{tag}'] = npy(L(an, b_))
        work = a - b_ - d
        out[f'fep_plain_{tag}'] = npy(fep_estimator(work))
        out[f'fep_kT_{tag}'] = npy(fep_estimator(work * 2.5, kT=2.5))
        out[f'fep_biased_{tag}'] = npy(fep_estimator(torch.stack([work, c], dim=1)))
        boot = torch.stack([work[torch.randint(0, N, (N,), generator=gen(s))] for s in range(4)])
        out[f'fep_vec_in_{tag}'] = npy(boot)
        out[f'fep_vec_{tag}'] = npy(fep_estimator(boot, vectorized=True))
        wts = torch.rand(4, N, generator=gen(9)).to(dt)
        wts = wts / wts.sum(dim=1, keepdim=True)
        out[f'fep_bayes_w_{tag}'] = npy(wts)
        out[f'fep_bayes_{tag}'] = npy(fep_estimator(work.expand(4, N), weights=wts, vectorized=True))
        bootb = torch.stack([torch.stack([work, c], dim=1)[torch.randint(0, N, (N,), generator=gen(s))]
                             for s in range(3)])
        out[f'fep_vecb_in_{tag}'] = npy(bootb)
        out[f'fep_vecb_{tag}'] = npy(fep_estimator(bootb, vectorized=True))
    torch.set_default_dtype(torch.float32)
    np.savez_compressed(os.path.join(OUT, 'loss.npz'), **out)


# -----------------------------------------------------------------------------
# 8. flow wrappers (PartialFlow / CenteredCentroidFlow / OrientedFlow) and the frame rotation
# -----------------------------------------------------------------------------

def gen_wrappers():
    import types
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import golden_util as gu
    from tfep.nn.flows.centroid import CenteredCentroidFlow
    from tfep.nn.flows.oriented import OrientedFlow
    from tfep.nn.flows.partial import PartialFlow
    from tfep.utils.geometry import reference_frame_rotation_matrix, get_axis_from_name
    ref_flows = types.SimpleNamespace(CenteredCentroidFlow=CenteredCentroidFlow, OrientedFlow=OrientedFlow,
                                      PartialFlow=PartialFlow)
    out = {}
    for i, (name, cfg) in enumerate(gu.wrapper_configs().items()):
        n_in = gu.wrapper_n_inner(cfg)
        x = 1.5 * torch.randn(96, cfg['n_points'] * cfg['dim'], generator=gen(4000 + i))

        def make(dt, cfg=cfg, n_in=n_in):
            if cfg.get('spline'):
                tr = NeuralSplineTransformer(x0=torch.full((n_in,), -8.0), xf=torch.full((n_in,), 8.0), n_bins=6)
            else:
                tr = AffineTransformer()
            inner = MAF(degrees_in=generate_degrees(n_in, order='ascending'), transformer=tr,
                        initialize_identity=False)
            return gu.build_wrapped(cfg, inner, ref_flows)
        run_flow(make, x, 500 + i, out, name, inverse=cfg['inverse'])
        # gradients of  sum(y * c) + sum(ldj)  in float64, c[b, j] = cos(b + 2 j)
        with f64():
            m = make(torch.float64)
            m.load_state_dict(to_double_sd({k[len(name) + 4:]: torch.from_numpy(v) for k, v in out.items()
                                            if k.startswith(name + '/sd/')}), strict=False)
            xg = x.double().requires_grad_(True)
            y, ldj = m(xg)
            c = torch.cos(torch.arange(y.shape[0]).unsqueeze(1) + 2.0 * torch.arange(y.shape[1]).unsqueeze(0))
            ((y * c).sum() + ldj.sum()).backward()
            out[f'{name}/gx_f64'] = npy(xg.grad)
            for k, prm in m.named_parameters():
                out[f'{name}/gp/{k}'] = npy(prm.grad)

    # frame rotation matrices on their own (float64), all axis / plane choices
    pos = torch.randn(64, 2, 3, generator=gen(4100), dtype=torch.float64)
    out['frame/axis_pos'], out['frame/plane_pos'] = npy(pos[:, 0]), npy(pos[:, 1])
    with f64():
        for axis in 'xyz':
            for plane_axis in 'xyz':
                if plane_axis == axis:
                    continue
                a, p = get_axis_from_name(axis), get_axis_from_name(plane_axis)
                for positive in (False, True):
                    r = reference_frame_rotation_matrix(pos[:, 0], pos[:, 1], a, p, project_on_positive_axis=positive)
                    out[f'frame/{axis}{plane_axis}{int(positive)}'] = npy(r)
    np.savez_compressed(os.path.join(OUT, 'wrappers.npz'), **out)


# -----------------------------------------------------------------------------
# 9. flip-invariant / mixed embeddings: standalone and in front of a MAF (forward, inverse, gradients)
# -----------------------------------------------------------------------------

def gen_embeddings():
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import golden_util as gu
    out = {}
    for i, (name, cfg) in enumerate(gu.embedding_configs().items()):
        torch.manual_seed(900 + i)
        emb = gu.build_embedding(cfg, sys.modules[__name__])
        x = torch.randn(24, cfg['n_features_in'], generator=gen(5000 + i))
        with f64():
            e64 = gu.build_embedding(cfg, sys.modules[__name__])
            e64.load_state_dict(to_double_sd(emb.state_dict()))
            out[f'emb/{name}/out_f64'] = npy(e64(x.double()))
        out[f'emb/{name}/x'] = npy(x)
        out[f'emb/{name}/out_f32'] = npy(emb(x))
        for k, v in emb.state_dict().items():
            out[f'emb/{name}/sd/{k}'] = npy(v)
        deg = torch.as_tensor(cfg['degrees_in'])
        out[f'emb/{name}/degrees_out'] = npy(emb.get_degrees_out(deg))

    for i, (name, cfg) in enumerate(gu.embedded_flow_configs().items()):
        x = torch.randn(40, len(cfg['degrees_in']), generator=gen(5100 + i)) * 0.8

        def make(dt, cfg=cfg):
            n_tr = sum(1 for d in cfg['degrees_in'] if d >= 0)
            if cfg['transformer'] == 'spline':
                tr = NeuralSplineTransformer(x0=torch.full((n_tr,), -4.0), xf=torch.full((n_tr,), 4.0), n_bins=8)
            else:
                tr = AffineTransformer()
            return SequentialFlow(MAF(degrees_in=torch.as_tensor(cfg['degrees_in']), transformer=tr,
                                      embedding=gu.build_embedding(cfg['embedding'], sys.modules[__name__]),
                                      initialize_identity=False))
        run_flow(make, x, 700 + i, out, name, inverse=True)
        with f64():
            m = make(torch.float64)
            m.load_state_dict(to_double_sd({k[len(name) + 4:]: torch.from_numpy(v) for k, v in out.items()
                                            if k.startswith(name + '/sd/')}), strict=False)
            xg = x.double().requires_grad_(True)
            y, ldj = m(xg)
            c = torch.cos(torch.arange(y.shape[0]).unsqueeze(1) + 2.0 * torch.arange(y.shape[1]).unsqueeze(0))
            ((y * c).sum() + ldj.sum()).backward()
            out[f'{name}/gx_f64'] = npy(xg.grad)
            for k, prm in m.named_parameters():
                out[f'{name}/gp/{k}'] = npy(prm.grad)
    np.savez_compressed(os.path.join(OUT, 'embeddings.npz'), **out)


# -----------------------------------------------------------------------------
# 10. bootstrap of fep_estimator (analysis/bootstrap.py)
# -----------------------------------------------------------------------------

def gen_bootstrap():
    from tfep.analysis.bootstrap import bootstrap
    out = {}
    g0 = gen(7000)
    work = torch.randn(600, generator=g0) * 1.5 + 0.3
    bias = torch.randn(600, generator=g0) * 0.5
    out['work'], out['bias'] = npy(work), npy(bias)
    cases = {
        'plain': dict(data='work', kw=dict(n_resamples=300)),
        'batched_cl80': dict(data='work', kw=dict(n_resamples=250, batch=64, confidence_level=0.8)),
        'sizes': dict(data='work', kw=dict(n_resamples=200, bootstrap_sample_size=[50, 600])),
        'sizes_first': dict(data='work', kw=dict(n_resamples=200, bootstrap_sample_size=[50, 300], take_first_only=True)),
        'biased': dict(data='biased', kw=dict(n_resamples=200, batch=50)),
    }
    for name, c in cases.items():
        data = work if c['data'] == 'work' else torch.stack([work, bias], dim=1)
        res = bootstrap(data, fep_estimator, generator=gen(7100), **c['kw'])
        res = res if isinstance(res, list) else [res]
        out[f'{name}/n'] = np.array(len(res))
        for i, r in enumerate(res):
            out[f'{name}/{i}'] = np.array([float(r['confidence_interval']['low']), float(r['confidence_interval']['high']),
                                           float(r['standard_deviation']), float(r['mean']), float(r['median'])])
    # Bayesian bootstrap: the Dirichlet weights are stored (drawn from the global RNG in the reference)
    torch.manual_seed(7200)
    weights = torch.distributions.Dirichlet(torch.ones(600)).sample((40,))
    out['bayes/weights'] = npy(weights)
    out['bayes/df'] = npy(fep_estimator(work.expand(40, 600), weights=weights, vectorized=True))
    # explicit resamples: indices and the per-resample estimates
    idx = torch.randint(0, 600, (30, 450), generator=gen(7300))
    out['explicit/idx'] = npy(idx)
    out['explicit/df_work'] = npy(fep_estimator(work[idx], vectorized=True))
    out['explicit/df_biased'] = npy(fep_estimator(torch.stack([work, bias], dim=1)[idx], vectorized=True))
    out['explicit/df_work_kT'] = npy(fep_estimator(work[idx], kT=0.7, vectorized=True))
    np.savez_compressed(os.path.join(OUT, 'bootstrap.npz'), **out)


def gen_logger():
    """A small log directory written by the reference TFEPLogger (data files only): tests/golden/logger_ref/, plus the
    inputs of every call and the reference's answers to a few reads in logger.npz."""
    import shutil
    import warnings
    from tfep.io.log import TFEPLogger
    root = os.path.join(OUT, 'logger_ref')
    shutil.rmtree(root, ignore_errors=True)
    n, bs = 9, 2
    loader = torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(n)), batch_size=bs, drop_last=False)
    log = TFEPLogger(save_dir_path=root, data_loader=loader)
    g0 = gen(8000)
    out = {'n': np.array(n), 'batch_size': np.array(bs)}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # training: epoch 0 batches 0, 1, 3, 4 (batch 2 never saved; the last batch has one sample), epoch 1 whole epoch
        for b in (0, 1, 3, 4):
            m = min(bs, n - b * bs)
            t = {'dataset_sample_index': torch.arange(b * bs, b * bs + m), 'potential': torch.randn(m, generator=g0),
                 'log_det_J': torch.randn(m, generator=g0)}
            if b == 3:
                t['potential'][0] = float('nan')
            for k, v in t.items():
                out[f'train/e0b{b}/{k}'] = npy(v)
            log.save_train_tensors(t, epoch_idx=0, batch_idx=b)
        t = {'dataset_sample_index': torch.randperm(n, generator=g0), 'potential': torch.randn(n, generator=g0)}
        for k, v in t.items():
            out[f'train/e1/{k}'] = npy(v)
        log.save_train_tensors(t, epoch_idx=1)
        # evaluation at step 7: three batches out of order, then an update of two stored samples
        for j, idx in enumerate(([4, 5, 6], [0, 1], [7, 8, 2, 3])):
            t = {'trajectory_sample_index': torch.tensor(idx), 'work': torch.randn(len(idx), generator=g0)}
            if j == 1:
                t['work'][1] = float('nan')
            for k, v in t.items():
                out[f'eval/s7c{j}/{k}'] = npy(v)
            log.save_eval_tensors(t, step_idx=7)
        t = {'trajectory_sample_index': torch.tensor([5, 0, 11]), 'work': torch.tensor([10.0, 20.0, 30.0])}
        for k, v in t.items():
            out[f'eval/s7upd/{k}'] = npy(v)
        log.save_eval_tensors(t, step_idx=7, update=True)
    reads = {
        'r_train_e0': log.read_train_tensors(epoch_idx=0, as_numpy=True),
        'r_train_e0_nonan': log.read_train_tensors(epoch_idx=0, remove_nans=True, as_numpy=True),
        'r_train_e0b3': log.read_train_tensors(epoch_idx=0, batch_idx=3, remove_nans='potential', as_numpy=True),
        'r_train_step6': log.read_train_tensors(step_idx=6, names=['potential'], as_numpy=True),
        'r_eval': log.read_eval_tensors(step_idx=7, as_numpy=True),
        'r_eval_nonan': log.read_eval_tensors(epoch_idx=1, batch_idx=2, remove_nans=True, as_numpy=True),
    }
    for rname, d in reads.items():
        for k, v in d.items():
            out[f'{rname}/{k}'] = np.asarray(v)
    np.savez_compressed(os.path.join(OUT, 'logger.npz'), **out)
    # (sorting rewrites the file: do it on a copy, the unsorted directory above stays the read fixture)
    sorted_root = os.path.join(OUT, 'logger_ref_sorted')
    shutil.rmtree(sorted_root, ignore_errors=True)
    shutil.copytree(root, sorted_root)
    TFEPLogger(save_dir_path=sorted_root).read_eval_tensors(step_idx=7, sort_by='trajectory_sample_index')


# -----------------------------------------------------------------------------
# 12. config 5: EGNN dynamics, radial bases, graph helpers, trace estimators of the continuous flow
# -----------------------------------------------------------------------------

def continuous_configs():
    """Small EGNN dynamics configurations (constructor arguments + batch); also imported by tests/golden_util.py."""
    return {
        # every pair inside the cutoff, two layers, tiny widths
        'tiny': dict(node_types=[0, 0, 1, 2, 1], r_cutoff=50.0, time_feat_dim=4, node_feat_dim=8, distance_feat_dim=6,
                     n_layers=2, speed_factor=1.0, batch=7, x_scale=1.0, seed=11),
        # a cutoff that prunes about half of the pairs; speed_factor != 1
        'cutoff': dict(node_types=[0, 1, 1, 0, 2, 2, 0], r_cutoff=1.6, time_feat_dim=3, node_feat_dim=16,
                       distance_feat_dim=8, n_layers=3, speed_factor=0.7, batch=9, x_scale=0.9, seed=12),
        # the default widths of EGNNDynamics (egnn.py:73-81) on 12 nodes, 4 layers
        'default': dict(node_types=[0, 1, 1, 0, 2, 2, 0, 3, 1, 0, 2, 1], r_cutoff=2.5, time_feat_dim=16, node_feat_dim=64,
                        distance_feat_dim=64, n_layers=4, speed_factor=1.0, batch=5, x_scale=1.2, seed=13),
        # a single node pair and one node type (degenerate sizes)
        'pair': dict(node_types=[0, 0], r_cutoff=10.0, time_feat_dim=2, node_feat_dim=4, distance_feat_dim=3,
                     n_layers=1, speed_factor=1.0, batch=3, x_scale=1.0, seed=14),
    }


def gen_continuous():
    from tfep.nn.dynamics.egnn import EGNNDynamics
    from tfep.nn.embeddings.radial import BehlerParrinelloRadialExpansion, GaussianBasisExpansion
    from tfep.nn.flows.continuous import ContinuousFlow
    from tfep.nn import graph as rgraph
    out = {}

    # ---- radial bases (embeddings/radial.py:90-130, 269-291)
    r = torch.cat([torch.tensor([0.0, 1e-3, 2.0, 2.0 + 1e-6, 3.5]), torch.rand(40, generator=gen(1)) * 2.4]).float()
    for dname, dt in (('f64', torch.float64), ('f32', torch.float32)):
        old = torch.get_default_dtype()
        torch.set_default_dtype(dt)
        try:
            gb = GaussianBasisExpansion.from_range(n_gaussians=5, max_mean=1.0, trainable_stds=True)
            bp = BehlerParrinelloRadialExpansion.from_range(r_cutoff=2.0, n_gaussians=6, max_mean=2.0, trainable_stds=True)
            bp2 = BehlerParrinelloRadialExpansion(r_cutoff=2.0, means=torch.tensor([0.1, 0.7, 1.9]),
                                                  stds=torch.tensor([0.3, 0.2, 0.5]), force_zero_after_cutoff=False)
            with torch.no_grad():
                out[f'radial/gauss_{dname}'] = npy(gb(r.to(dt)))
                out[f'radial/bp_{dname}'] = npy(bp(r.to(dt)))
                out[f'radial/bp_nozero_{dname}'] = npy(bp2(r.to(dt)))
            if dname == 'f64':
                out['radial/gauss_means'], out['radial/gauss_log_gammas'] = npy(gb._means), npy(gb._log_gammas)
                out['radial/bp_means'], out['radial/bp_log_gammas'] = npy(bp._means), npy(bp._log_gammas)
                out['radial/bp2_means'], out['radial/bp2_log_gammas'] = npy(bp2._means), npy(bp2._log_gammas)
        finally:
            torch.set_default_dtype(old)
    out['radial/r'] = npy(r)

    # ---- graph helpers (graph.py:119-316)
    for n in (1, 2, 4):
        out[f'graph/edges_n{n}_b1'] = npy(rgraph.get_all_edges(1, n)).astype(np.int64)
        out[f'graph/edges_n{n}_b3'] = npy(rgraph.get_all_edges(3, n)).astype(np.int64)
    mask = torch.tensor([[0., 1, 0], [1, 0, 1], [1, 1, 0]])
    out['graph/mask'] = npy(mask)
    out['graph/edges_masked_b2'] = npy(rgraph.get_all_edges(2, 3, mask)).astype(np.int64)
    xg = torch.randn(8, 3, generator=gen(2), dtype=torch.float64)
    eg = rgraph.get_all_edges(2, 4)
    for norm in (False, True):
        d, v = rgraph.compute_edge_distances(xg, eg, normalize_directions=norm)
        out[f'graph/dist_norm{int(norm)}'], out[f'graph/dir_norm{int(norm)}'] = npy(d), npy(v)
    d, v = rgraph.compute_edge_distances(xg, eg, normalize_directions=True)
    pe, pd, pv = rgraph.prune_long_edges(1.5, eg, d, v)
    out['graph/x'], out['graph/pruned_edges'], out['graph/pruned_dist'], out['graph/pruned_dir'] = \
        npy(xg), npy(pe).astype(np.int64), npy(pd), npy(pv)
    data = torch.randn(len(d), 3, generator=gen(3), dtype=torch.float64)
    out['graph/seg_data'] = npy(data)
    out['graph/seg_sum'] = npy(rgraph.unsorted_segment_sum(data, eg[1], 8))

    # ---- EGNN dynamics, ODE function (velocity, trace, regularisation) with stored eps
    for name, cfg in continuous_configs().items():
        kw = {k: cfg[k] for k in ('node_types', 'r_cutoff', 'time_feat_dim', 'node_feat_dim', 'distance_feat_dim',
                                  'n_layers', 'speed_factor')}
        n_nodes, B = len(cfg['node_types']), cfg['batch']
        g = gen(cfg['seed'])
        x32 = (torch.randn(B, 3 * n_nodes, generator=g) * cfg['x_scale']).float()
        t32 = torch.rand(1, generator=g).float()
        eps32 = torch.randn(3, B, 3 * n_nodes, generator=g).float()
        torch.manual_seed(cfg['seed'])
        dyn32 = EGNNDynamics(initialize_identity=False, **kw)
        # spread every parameter (biases are small, the log-gammas identical at init) -- still float32-rounded
        with torch.no_grad():
            for prm in dyn32.parameters():
                prm.add_(0.05 * torch.randn(prm.shape, generator=g))
        sd32 = dyn32.state_dict()
        out.update({f'{name}/sd/{k}': npy(v) for k, v in sd32.items()})
        out[f'{name}/x'], out[f'{name}/t'], out[f'{name}/eps'] = npy(x32), npy(t32), npy(eps32)
        for dname, dt in (('f64', torch.float64), ('f32', torch.float32)):
            old = torch.get_default_dtype()
            torch.set_default_dtype(dt)
            try:
                dyn = EGNNDynamics(initialize_identity=False, **kw)
                dyn.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd32.items()})
                x, t, eps = x32.to(dt), t32.to(dt), eps32.to(dt)
                with torch.no_grad():
                    out[f'{name}/vel_{dname}'] = npy(dyn(t[0], x))
                    if dname == 'f64':
                        # node embedding and the state after every layer (egnn.py:158-176)
                        h = dyn._create_node_embedding(t[0], B)
                        out[f'{name}/h0_f64'] = npy(h)
                        pos = x.view(B * n_nodes, 3)
                        edges = dyn.get_edges(B)
                        for li in range(cfg['n_layers']):
                            h, pos = dyn._modules['graph_layer_' + str(li)](h, pos, edges)
                            out[f'{name}/h{li + 1}_f64'], out[f'{name}/pos{li + 1}_f64'] = npy(h), npy(pos)
                # the integrands of the ODE function (continuous.py:231-278) with the stored eps
                for est, n_hut, reg in (('hutchinson', 1, True), ('hutchinson', 3, True), ('hutchinson', 1, False),
                                        ('exact', 1, True), ('exact', 1, False)):
                    flow = ContinuousFlow(dyn, trace_estimator=est, n_hutchinson_samples=n_hut, regularization=reg,
                                          requires_backward=False)
                    f = flow.ode_func
                    if est == 'hutchinson':
                        f._eps = eps[:n_hut]
                    else:
                        f._cached_eye = torch.eye(x.shape[1])
                    xx = x.clone()
                    state = (xx, xx.new_zeros(B), xx.new_zeros(B)) if reg else (xx, xx.new_zeros(B))
                    res = f(t[0], state)
                    key = f'{name}/{est}{n_hut}_{"reg" if reg else "noreg"}'
                    out[f'{key}/vel_{dname}'] = npy(res[0])
                    out[f'{key}/trace_{dname}'] = npy(res[1])
                    if reg:
                        out[f'{key}/reg_{dname}'] = npy(res[2])
                if dname == 'f64':
                    # the full Jacobian d vel / d x of every sample (pins JVP / VJP products of any direction)
                    jac = torch.stack([torch.autograd.functional.jacobian(lambda z: dyn(t[0], z[None])[0], x[b])
                                       for b in range(B)])
                    out[f'{name}/jacobian_f64'] = npy(jac)
            finally:
                torch.set_default_dtype(old)

    # ---- identity initialisation: zero velocity (egnn.py:136-138)
    torch.manual_seed(0)
    dyn = EGNNDynamics(node_types=[0, 1, 0], r_cutoff=5.0, time_feat_dim=2, node_feat_dim=4, distance_feat_dim=3, n_layers=2)
    xi = torch.randn(4, 9, generator=gen(5))
    with torch.no_grad():
        out['identity/vel_max_abs'] = np.asarray(float(dyn(torch.tensor(0.3), xi).abs().max()))
    out.update({f'identity/sd/{k}': npy(v) for k, v in dyn.state_dict().items()})
    np.savez_compressed(os.path.join(OUT, 'continuous.npz'), **out)


def gen_continuous_grads():
    """Gradients through the ODE function of the continuous flow (reference continuous.py:231-278 with ``create_graph``:
    what ``loss.backward()`` of a training step differentiates at every solver stage).  For the dynamics of
    ``continuous.npz`` (same parameters): integrands (vel, trace, reg) at (t, x) with the Hutchinson (2 samples) and the
    exact estimator, loss = <a, vel> + <b, trace> + <c, reg> with stored random a, b, c; d loss / d x and d loss / d(every
    parameter) in float64 -- second derivatives of the dynamics (the trace is a first derivative already)."""
    from tfep.nn.dynamics.egnn import EGNNDynamics
    from tfep.nn.flows.continuous import ContinuousFlow
    base = np.load(os.path.join(OUT, 'continuous.npz'))
    out = {}
    for name, cfg in continuous_configs().items():
        if name == 'default':
            continue                                    # 36 coordinates x 64 features: covered by the smaller ones
        kw = {k: cfg[k] for k in ('node_types', 'r_cutoff', 'time_feat_dim', 'node_feat_dim', 'distance_feat_dim',
                                  'n_layers', 'speed_factor')}
        x32, t32, eps32 = torch.from_numpy(base[f'{name}/x']), torch.from_numpy(base[f'{name}/t']), torch.from_numpy(base[f'{name}/eps'])
        B, D = x32.shape
        g = gen(900 + cfg['seed'])
        a, b, c = torch.randn(B, D, generator=g), torch.randn(B, generator=g), torch.randn(B, generator=g) * 0.1
        out[f'{name}/a'], out[f'{name}/b'], out[f'{name}/c'] = npy(a), npy(b), npy(c)
        with f64():
            dyn = EGNNDynamics(initialize_identity=False, **kw)
            sd = {k[len(name) + 4:]: torch.from_numpy(base[k]) for k in base.files if k.startswith(f'{name}/sd/')}
            dyn.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()})
            for est, n_hut in (('hutchinson', 2), ('exact', 1)):
                flow = ContinuousFlow(dyn, trace_estimator=est, n_hutchinson_samples=n_hut, regularization=True,
                                      requires_backward=True)
                f = flow.ode_func
                if est == 'hutchinson':
                    f._eps = eps32[:n_hut].double()
                else:
                    f._cached_eye = torch.eye(D)
                for prm in dyn.parameters():
                    prm.grad = None
                x = x32.double().requires_grad_(True)
                vel, trace, reg = f(t32.double()[0], (x, x.new_zeros(B), x.new_zeros(B)))
                loss = (a.double() * vel).sum() + (b.double() * trace).sum() + (c.double() * reg).sum()
                loss.backward()
                key = f'{name}/{est}'
                out[f'{key}/loss'] = npy(loss)
                out[f'{key}/gx'] = npy(x.grad)
                for k, prm in dyn.named_parameters():
                    out[f'{key}/grad/{k}'] = npy(prm.grad if prm.grad is not None else torch.zeros_like(prm))
    np.savez_compressed(os.path.join(OUT, 'continuous_grads.npz'), **out)


# -----------------------------------------------------------------------------
# 13. PCAWhitenedFlow around a MAF (reference flows/pca.py): forward / inverse, blackened or not, gradients
# -----------------------------------------------------------------------------

def gen_pca():
    sys.path.insert(0, os.path.join(os.path.dirname(OUT)))
    import golden_util as gu
    from tfep.nn.flows.pca import PCAWhitenedFlow
    out = {}
    for i, (name, cfg) in enumerate(gu.pca_configs().items()):
        D = cfg['D']
        data = gu.pca_data(cfg)                                   # correlated, off-centre samples the PCA is estimated from
        x = data[:cfg['batch']].clone()

        def make(dt, cfg=cfg, D=D):
            if cfg['spline']:
                tr = NeuralSplineTransformer(x0=torch.full((D,), -9.0), xf=torch.full((D,), 9.0), n_bins=5)
            else:
                tr = AffineTransformer()
            inner = MAF(degrees_in=generate_degrees(D, order=cfg['order']), transformer=tr, initialize_identity=False)
            return PCAWhitenedFlow(inner, data.to(dt), blacken=cfg['blacken'])
        run_flow(make, x, 900 + i, out, name, inverse=True)
        with f64():
            m = make(torch.float64)
            m.load_state_dict(to_double_sd({k[len(name) + 4:]: torch.from_numpy(v) for k, v in out.items()
                                            if k.startswith(name + '/sd/')}), strict=False)
            xg = x.double().requires_grad_(True)
            y, ldj = m(xg)
            c = torch.cos(torch.arange(y.shape[0]).unsqueeze(1) + 2.0 * torch.arange(y.shape[1]).unsqueeze(0))
            ((y * c).sum() + ldj.sum()).backward()
            out[f'{name}/gx_f64'] = npy(xg.grad)
            for k, prm in m.named_parameters():
                out[f'{name}/gp/{k}'] = npy(prm.grad)
    np.savez_compressed(os.path.join(OUT, 'pca.npz'), **out)


if __name__ == '__main__':
    torch.set_num_threads(4)
    if len(sys.argv) > 1:
        for fn in sys.argv[1:]:
            globals()['gen_' + fn]()
        sys.exit(0)
    gen_degrees()
    gen_masked()
    gen_made()
    gen_transformers()
    gen_flows()
    gen_grads()
    gen_inv_grads()
    gen_grads_wide()
    gen_loss()
    gen_wrappers()
    gen_embeddings()
    gen_bootstrap()
    gen_logger()
    gen_continuous()
    gen_continuous_grads()
    gen_pca()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
