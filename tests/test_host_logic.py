"""CPU tests of the host side of tfep_amd: constructors, degrees / masks, state_dict schema,
error conventions, the C ABI symbol table.  No kernel is launched (no GPU here)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import golden_util as gu
from tfep_amd import _lib
from tfep_amd.nn import masked
from tfep_amd.nn.conditioners import MADE, generate_degrees
from tfep_amd.nn.flows import MAF, AutoregressiveFlow, SequentialFlow
from tfep_amd.nn.transformers import (AffineTransformer, MixedTransformer, MoebiusTransformer,
                                      NeuralSplineTransformer)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------ C ABI

def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'tfep_hip.h')).read()
    declared = set(re.findall(r'\b(tfep_[a-z0-9_]+)\s*\(', header))
    declared -= {'tfep_param_layout', 'tfep_spline_desc', 'tfep_gemm_desc'}
    assert declared, 'no declarations parsed'
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f'{name} declared in tfep_hip.h but not exported'
    # the ctypes binding covers exactly the declared entry points
    assert set(_lib.EXPORTED_SYMBOLS) == declared
    assert _lib.load().tfep_hip_abi_version() == _lib.ABI_VERSION == 8


def test_no_cpu_fallback():
    D = 4
    flow = SequentialFlow(MAF(generate_degrees(D)))
    with pytest.raises(_lib.TfepHipError, match='no CPU fallback'):
        flow(torch.randn(2, D))
    with pytest.raises(_lib.TfepHipError, match='no CPU fallback'):
        AffineTransformer()(torch.randn(2, D), torch.randn(2, 2 * D))


# ------------------------------------------------------------------ degrees / masks (known-answer tables)

def test_generate_degrees_table():
    for case in gu.load_json('degrees.json')['generate_degrees']:
        assert generate_degrees(**case['kwargs']).tolist() == case['expected'], case['kwargs']
    with pytest.raises(ValueError, match="Accepted string values"):
        generate_degrees(4, order='sideways')


def test_degrees_hidden_table():
    for case in gu.load_json('degrees.json')['degrees_hidden']:
        got = MADE._get_degrees_hidden(torch.tensor(case['degrees_in']), torch.tensor(case['degrees_out']),
                                       case['hidden_layers'])
        assert [g.tolist() for g in got] == case['expected']
    with pytest.raises(ValueError, match='too small'):
        MADE(degrees_in=[0, 1, 2, 3, 4], degrees_out=[0, 1, 2, 3, 4], hidden_layers=[3])
    with pytest.raises(ValueError, match='ignored'):
        MADE(degrees_in=[0, 1, 2], degrees_out=[0, 1, 2], hidden_layers=[[0, 1, 2]])


def test_mask_table_and_made_masks():
    j = gu.load_json('degrees.json')
    for case in j['masks']:
        m = masked.create_autoregressive_mask(np.array(case['degrees_in']), np.array(case['degrees_out']),
                                              strictly_less=case['strictly_less'], transpose=case['transpose'])
        assert m.dtype == torch.get_default_dtype()
        assert m.int().tolist() == case['expected']
    mm = j['made_masks']
    made = MADE(mm['degrees_in'], mm['degrees_out'], mm['hidden_layers'])
    assert [l.mask.int().tolist() for l in made.layers[::2]] == mm['expected']
    for case in j['mask_nnz'][:2]:
        deg = generate_degrees(case['D'], case['order'])
        made = MADE(deg, deg.tile((case['P'],)), 2, weight_norm=False)
        assert [list(l.mask.shape) for l in made.layers[::2]] == case['shapes']
        assert [int(l.mask.sum()) for l in made.layers[::2]] == case['nnz']


# ------------------------------------------------------------------ state_dict schema == reference's

@pytest.mark.parametrize('name', ['cfg1', 'rq4', 'cond', 'circ', 'moeb', 'mixflow', 'ident'])
def test_state_dict_schema_matches_reference(name):
    g = gu.load('flows.npz')
    flow = gu.build_flow(name, g, device=None)      # asserts keys / shapes / dtypes both ways
    n_par = g[f'{name}/n_parameters'] if f'{name}/n_parameters' in g.files else None
    if n_par is not None:
        assert int(flow.n_parameters()) == int(n_par)
    if name in ('cfg1', 'cond'):
        sd = flow.state_dict()
        for k in sd:
            if k.endswith('.mask'):
                shape = tuple(g[f'{name}/sdmaskshape/{k}'])
                bits = np.unpackbits(g[f'{name}/sdmask/{k}'])[:shape[0] * shape[1]].reshape(shape)
                assert np.array_equal(bits.astype(bool), sd[k].numpy() != 0), k


def test_masked_linear_module_schema():
    mask = torch.tril(torch.ones(5, 8))
    mask[2] = 0
    lin = masked.MaskedLinear(8, 5, mask=mask)
    assert torch.all(lin.weight[mask == 0] == 0)
    assert int(lin.n_parameters()) == int(mask.sum()) + 5
    lin = masked.masked_weight_norm(lin)
    assert sorted(lin.state_dict().keys()) == ['bias', 'mask', 'weight_g', 'weight_v']
    assert lin.weight_g.shape == (5, 1) and lin.weight_v.shape == (5, 8)
    with pytest.raises(RuntimeError, match='two weight_norm'):
        masked.masked_weight_norm(lin)
    # weight norm init: g = row norm of the masked weight (masked.py:391)
    np.testing.assert_allclose(lin.weight_g[:, 0].detach().numpy(),
                               np.linalg.norm(lin.weight_v.detach().numpy(), axis=1), rtol=1e-6)
    g = gu.load('masked_linear.npz')
    assert sorted(lin.state_dict().keys()) == list(g['wn_state_keys'])
    masked.remove_masked_weight_norm(lin)
    assert 'weight' in lin._parameters and 'weight_g' not in lin._parameters
    assert torch.all(torch.isfinite(lin.weight)) and torch.all(lin.weight[2] == 0)


# ------------------------------------------------------------------ constructor error conventions

def test_error_conventions():
    with pytest.raises(ValueError, match='consecutive values'):
        MAF(degrees_in=[0, 2, 3])
    with pytest.raises(ValueError, match='consecutive values'):
        MAF(degrees_in=[1, 2, 3])
    with pytest.raises(ValueError, match='0 <= i < n_features_in'):
        AutoregressiveFlow(3, [[0], [5]], MADE([0, 1, 2], [0, 1, 2] * 2), AffineTransformer())
    x0 = torch.zeros(3)
    with pytest.raises(ValueError, match='circular spline with learnable limits'):
        NeuralSplineTransformer(x0, x0 + 1, 4, circular=True, learn_lower_bound=True)
    with pytest.raises(ValueError, match='x0==y0 and xf==yf'):
        NeuralSplineTransformer(x0, x0 + 1, 4, y0=x0 + 0.5, circular=True)
    with pytest.raises(ValueError, match='minimum bin size'):
        NeuralSplineTransformer(x0, x0 + 1, 4, min_bin_size=0.0)
    with pytest.raises(ValueError, match='minimum slope'):
        NeuralSplineTransformer(x0, x0 + 1, 4, min_slope=1.0)
    with pytest.raises(ValueError, match='greater than 1'):
        MixedTransformer([AffineTransformer()], [[0, 1]])
    with pytest.raises(ValueError, match='must equal'):
        MixedTransformer([AffineTransformer(), AffineTransformer()], [[0, 1]])
    with pytest.raises(ValueError, match='only if x0=y0'):
        NeuralSplineTransformer(x0, x0 + 1, 4, y0=x0 + 0.5).get_identity_parameters(3)


@pytest.mark.parametrize('circular,identity,ll,lu', gu.SPLINE_VARIANTS)
def test_spline_parameter_count_and_degrees(circular, identity, ll, lu):
    K, D = 5, 3
    t = NeuralSplineTransformer(torch.zeros(D), torch.ones(D), K, circular=circular,
                                identity_boundary_slopes=identity, learn_lower_bound=ll, learn_upper_bound=lu)
    n = (3 * K + 1) + (int(circular) - 2) * int(identity) + int(ll) + int(lu)
    assert t.n_parameters_per_feature == n
    assert len(t.get_identity_parameters(D)) == n * D
    assert t.get_degrees_out(torch.arange(D)).tolist() == list(range(D)) * n
    # the library agrees with the module (no launch involved)
    desc = _lib.SplineDesc(0, 0, 0, 0, K, int(circular), int(identity), int(ll), int(lu), 1e-4, 1e-4)
    assert _lib.load().tfep_spline_n_parameters_per_feature(ctypes.byref(desc)) == n


def test_identity_initialisation_sets_conditioner_output():
    D = 5
    maf = MAF(generate_degrees(D), transformer=NeuralSplineTransformer(torch.full((D,), -2.), torch.full((D,), 2.), 4))
    last = maf._conditioner.layers[-1]
    assert torch.all(last.weight_g == 0) and torch.all(last.bias == 0)
    maf = MAF(generate_degrees(D), weight_norm=False)
    assert torch.all(maf._conditioner.layers[-1]._parameters['weight'] == 0)
    assert MoebiusTransformer(2).get_identity_parameters(6).shape == (6,)


def test_inverse_masks_and_indices_buffers():
    maf = MAF(degrees_in=[-1, 0, 0, 1, -1, 2])
    assert maf._fixed_indices.tolist() == [0, 4]
    assert maf._transformer_indices.tolist() == [1, 2, 3, 5]
    assert maf._inverse_masks.int().tolist() == [[0, 1, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 1]]
    maf = MAF(degrees_in=[0, 1, 2])
    assert maf._transformer_indices.numel() == 0 and not maf.has_fixed_indices


@pytest.mark.parametrize('hidden', [2, [20, 17]])
def test_load_state_dict_rederives_degrees_from_the_loaded_buffers(hidden):
    """A checkpoint saved from a model with another (random) degree order replaces the masks; the host copy of the
    degrees that drives the execution plans / blocked inverse must follow (reference: everything derives from buffers)."""
    def build(seed, order='random'):
        torch.manual_seed(seed)
        return MAF(generate_degrees(7, order, conditioning_indices=[2]), transformer=AffineTransformer(),
                   hidden_layers=hidden)
    a, b = build(1), build(5)
    assert not torch.equal(a._conditioner._degrees[0], b._conditioner._degrees[0])
    b.load_state_dict(a.state_dict(), strict=True)
    assert b._conditioner._degrees_stale
    b._sync_conditioner()
    for da, db in zip(a._conditioner._degrees, b._conditioner._degrees):
        assert torch.equal(da, db)
    assert b._conditioner._degrees_ok and not b._conditioner._degrees_stale
    assert b._blocked_ok()
    # masks that no degree assignment reproduces: degree shortcuts off, nothing raises
    c = build(5)
    c._conditioner.layers[2].mask[0, :] = 1.0 - c._conditioner.layers[2].mask[0, :]
    c._conditioner.invalidate_plan()
    c._sync_conditioner()
    assert not c._conditioner._degrees_ok and not c._blocked_ok()


def test_custom_torch_ops_are_registered():
    """``torch.ops.tfep.*`` exist with schemas after importing the package's op module (no GPU needed to look)."""
    import tfep_amd.torch_ops as to
    for name in to.OPS:
        op = getattr(torch.ops.tfep, name)
        assert 'Tensor' in str(op.default._schema), name
    s = str(torch.ops.tfep.spline_forward.default._schema)
    assert 'n_bins' in s and 'circular' in s and '-> (Tensor, Tensor)' in s


def test_pca_whitened_flow_statistics_on_the_host():
    """PCAWhitenedFlow estimates its matrices at construction with torch (reference pca.py:52-76): W^T cov W = 1,
    B = W^-1, log-det = -sum(log singular values); a CPU tensor fails loudly instead of taking a CPU path."""
    import golden_util as gu
    from tfep_amd.nn.flows import PCAWhitenedFlow

    class Inner(torch.nn.Module):
        def forward(self, x):
            return x, torch.zeros(len(x))

        def n_parameters(self):
            return 3
    data = gu.pca_data(dict(D=7, n_data=300, seed=2)).double()
    flow = PCAWhitenedFlow(Inner(), data, blacken=False)
    assert list(flow.state_dict()) == ['mean', 'whitening_matrix', 'blackening_matrix', 'whitening_log_det_J']
    xc = data - data.mean(0)
    cov = xc.t() @ xc / (len(data) - 1)
    w, b = flow.whitening_matrix, flow.blackening_matrix
    assert torch.allclose(w.t() @ cov @ w, torch.eye(7, dtype=torch.float64), atol=1e-9)
    assert torch.allclose(w @ b, torch.eye(7, dtype=torch.float64), atol=1e-9)
    assert abs(float(flow.whitening_log_det_J) + 0.5 * float(torch.logdet(cov))) < 1e-9
    assert flow.n_parameters() == 3
    with pytest.raises(Exception):
        flow(data.float())


def test_which_layers_have_a_fused_kernel_and_a_blocked_inverse():
    """Host-side rules (no GPU): the fused output-GEMM + transformer kernel exists for the affine transformer, for RQ splines
    of 8 / 5 / 4 bins up to 25 parameters per feature, and for a mixed transformer whose members all qualify (the plain
    volume-preserving shift counts as an affine group); the blocked inverse takes mixed transformers of affine / spline /
    plain-shift members."""
    from tfep_amd.nn.conditioners import generate_degrees
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import (AffineTransformer, MixedTransformer, MoebiusTransformer, NeuralSplineTransformer,
                                          VolumePreservingShiftTransformer)

    def spline(n, K, **kw):
        return NeuralSplineTransformer(torch.zeros(n), torch.ones(n), K, **kw)

    def layer(tr, D=6):
        return MAF(generate_degrees(D, 'ascending'), transformer=tr, initialize_identity=False)
    assert layer(AffineTransformer())._fused_kind() == 0
    for K, kw, fused in ((8, {}, True), (5, dict(identity_boundary_slopes=True), True), (4, dict(circular=True), True),
                         (5, dict(identity_boundary_slopes=True, learn_lower_bound=True, learn_upper_bound=True), True),
                         (8, dict(identity_boundary_slopes=True, learn_upper_bound=True), True),       # 24 parameters
                         (8, dict(learn_upper_bound=True), True),                                       # 26
                         (8, dict(learn_lower_bound=True, learn_upper_bound=True), True),               # 27: the widest layout
                         (6, {}, False), (3, {}, False)):
        lay = layer(spline(6, K, **kw))
        assert (lay._fused_kind() == 1) == fused, (K, kw)
        assert lay._blocked_ok()
    lay.fused = False
    assert lay._fused_kind() is None
    # a plain shift on its own keeps the generic forward; as a member of a mixed transformer it is an affine group
    assert layer(VolumePreservingShiftTransformer())._fused_kind() is None
    mixed = MixedTransformer([spline(2, 5), spline(2, 5, circular=True), VolumePreservingShiftTransformer()], [[0, 1], [2, 3], [4, 5]])
    lay = layer(mixed)
    assert lay._fused_kind() == 2 and lay._blocked_ok() and lay._fused_inverse_supported(2)
    mixed = MixedTransformer([spline(3, 5), AffineTransformer()], [[0, 1, 2], [3, 4, 5]])
    lay = layer(mixed)
    assert lay._fused_kind() == 2 and lay._blocked_ok() and not lay._fused_inverse_supported(2)     # per-step launches
    periodic_shift = VolumePreservingShiftTransformer(periodic_indices=torch.tensor([0]), periodic_limits=torch.tensor([0.0, 1.0]))
    lay = layer(MixedTransformer([spline(3, 5), periodic_shift], [[0, 1, 2], [3, 4, 5]]))
    assert lay._fused_kind() is None and not lay._blocked_ok()
    lay = layer(MixedTransformer([spline(2, 5), MoebiusTransformer(dimension=2)], [[0, 1], [2, 3, 4, 5]]))
    assert lay._fused_kind() is None and not lay._blocked_ok()


def test_xcd_balanced_tile_list_covers_the_live_tiles_once_and_evenly():
    """``ops.xcd_balanced_tile_list`` (launch order of a block-sparse GEMM, ``tile_list`` of ``tfep_gemm_desc``): every live
    tile exactly once, none of the dead ones, groups of 32 dealt round-robin to the 8 XCD positions -- a triangular live
    region (the grad_weight of a MADE layer) gives every XCD the same number of tiles to within one group."""
    import torch
    from tfep_amd import ops
    for M, N in [(59, 59), (293, 38), (1, 1), (7, 3), (64, 5)]:
        live = torch.tril(torch.ones(M, N)).bool() if M > 1 else torch.ones(M, N).bool()
        tl = ops.xcd_balanced_tile_list(live)
        assert tl.dtype == torch.int32 and tl.shape[1] == 2 and tl.shape[0] % (8 * 32) == 0
        listed = tl[tl[:, 0] >= 0]
        assert bool((tl[tl[:, 0] < 0] == -1).all())
        assert listed.shape[0] == int(live.sum())
        assert {tuple(t) for t in listed.tolist()} == {tuple(t) for t in torch.nonzero(live).tolist()}
        per_xcd = [int((tl[x::8, 0] >= 0).sum()) for x in range(8)]
        assert max(per_xcd) - min(per_xcd) <= 32, (M, N, per_xcd)
    assert int((ops.xcd_balanced_tile_list(torch.zeros(4, 4).bool())[:, 0] >= 0).sum()) == 0
