"""Oracle (numpy): autoregressive / MAF / sequential flows.

TEST INFRASTRUCTURE ONLY -- see ``oracle/__init__.py``.
Reference files restated here: ``tfep/nn/flows/{autoregressive,maf,sequential}.py``.

A MAF layer is described by a plain dict::

    {'degrees_in':  int array (D,)           # -1 = conditioning feature
     'transformer': {'type': 'affine' | 'spline' | 'moebius' | 'volpres' | 'mixed', ...},
     'embedding':   None | {'type': 'periodic', 'limits': (lo, hi),
                            'periodic_indices': [...], 'nonperiodic_indices': [...]},
     'made':        [ {bias, mask, weight_g, weight_v | weight}, ... ]}
"""
import numpy as np

from . import made as _made
from . import transformers as _tr


def make_transformer(spec):
    """Return ``(forward_fn, inverse_fn, n_parameters_fn)`` for a transformer spec."""
    t = spec['type']
    if t == 'affine':
        return _tr.affine_forward, _tr.affine_inverse
    if t == 'volpres':
        kw = dict(periodic_indices=spec.get('periodic_indices'),
                  periodic_limits=spec.get('periodic_limits'))
        return (lambda x, p: _tr.volume_preserving_forward(x, p, **kw),
                lambda y, p: _tr.volume_preserving_inverse(y, p, **kw))
    if t == 'spline':
        kw = {k: v for k, v in spec.items() if k not in ('type', 'x0', 'xf', 'n_bins')}
        x0, xf, nb = spec['x0'], spec['xf'], spec['n_bins']
        return (lambda x, p: _tr.spline_forward(x, p, x0, xf, nb, **kw),
                lambda y, p: _tr.spline_inverse(y, p, x0, xf, nb, **kw))
    if t == 'moebius':
        kw = dict(dimension=spec['dimension'], max_radius=spec.get('max_radius', 0.99),
                  unit_sphere=spec.get('unit_sphere', False))
        return (lambda x, p: _tr.moebius_forward(x, p, **kw),
                lambda y, p: _tr.moebius_inverse(y, p, **kw))
    if t == 'mixed':
        subs = [make_transformer(s) for s in spec['transformers']]
        ind = [np.asarray(i) for i in spec['indices']]
        lens = spec['par_lengths']
        return (lambda x, p: _tr.mixed_run(x, p, subs, ind, lens, inverse=False),
                lambda y, p: _tr.mixed_run(y, p, subs, ind, lens, inverse=True))
    raise ValueError(t)


def _conditioner(x, layer):
    """_EmbeddedMADE.forward.  Ref: flows/maf.py:191-194."""
    emb = layer.get('embedding')
    if emb is not None:
        x = _tr.periodic_embedding(x, emb['limits'], np.asarray(emb['periodic_indices'], dtype=np.int64),
                                   np.asarray(emb['nonperiodic_indices'], dtype=np.int64))
    return _made.made_forward(x, layer['made'])


def _indices(layer):
    """transformer / fixed indices of a MAF layer.  Ref: flows/maf.py:152-153,
    flows/autoregressive.py:105-120."""
    deg = np.asarray(layer['degrees_in'])
    groups = [np.nonzero(deg == k)[0] for k in range(int(deg.max()) + 1)]
    tr_idx = np.sort(np.concatenate(groups))
    fixed = np.setdiff1d(np.arange(len(deg)), tr_idx)
    return groups, tr_idx, fixed


def maf_forward(x, layer):
    """AutoregressiveFlow.forward.  Ref: flows/autoregressive.py:144-177."""
    fwd, _ = make_transformer(layer['transformer'])
    params = _conditioner(x, layer)
    _, tr_idx, fixed = _indices(layer)
    if len(fixed) > 0:
        y = np.empty_like(x)
        y[:, fixed] = x[:, fixed]
        y[:, tr_idx], ldj = fwd(x[:, tr_idx], params)
    else:
        y, ldj = fwd(x, params)
    return y, ldj


def maf_inverse(y, layer):
    """AutoregressiveFlow.inverse: one full pass per degree.  Ref: flows/autoregressive.py:179-229."""
    _, inv = make_transformer(layer['transformer'])
    groups, tr_idx, fixed = _indices(layer)
    x = np.zeros_like(y)
    if len(fixed) > 0:
        x[:, fixed] = y[:, fixed]
        y_t = y[:, tr_idx]
    else:
        y_t = y
    pos = {int(f): i for i, f in enumerate(tr_idx)}
    ldj = None
    for grp in groups:
        params = _conditioner(x.copy(), layer)
        x_tmp, ldj = inv(y_t, params)                         # last pass' ldj is the total (:221)
        x[:, grp] = x_tmp[:, [pos[int(g)] for g in grp]]
    return x, ldj


def sequential_forward(x, layers):
    """SequentialFlow._pass(inverse=False).  Ref: flows/sequential.py:50-68."""
    ldj = np.zeros(x.shape[0], dtype=x.dtype)
    for layer in layers:
        x, l = maf_forward(x, layer)
        ldj = ldj + l
    return x, ldj


def sequential_inverse(y, layers):
    """SequentialFlow._pass(inverse=True): layers reversed.  Ref: flows/sequential.py:55-57."""
    ldj = np.zeros(y.shape[0], dtype=y.dtype)
    for layer in reversed(layers):
        y, l = maf_inverse(y, layer)
        ldj = ldj + l
    return y, ldj
