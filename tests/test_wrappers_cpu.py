"""Host logic of the flow wrappers (no GPU): frame rotation matrices against the reference's, index
bookkeeping and constructor errors (reference tests/nn/flows/test_oriented.py, test_centroid.py,
tests/utils/test_geometry.py)."""
import numpy as np
import pytest
import torch

import golden_util as gu
from tfep_amd.nn.flows import CenteredCentroidFlow, OrientedFlow
from tfep_amd.utils.geometry import batchwise_rotate, get_axis_from_name, reference_frame_rotation_matrix
from tfep_amd.utils.misc import atom_to_flattened, atom_to_flattened_indices, flattened_to_atom


@pytest.mark.parametrize('positive', [False, True])
@pytest.mark.parametrize('axis,plane_axis', [(a, p) for a in 'xyz' for p in 'xyz' if a != p])
def test_frame_rotation_matches_reference(axis, plane_axis, positive):
    g = gu.load('wrappers.npz')
    a_pos = torch.from_numpy(g['frame/axis_pos'])
    p_pos = torch.from_numpy(g['frame/plane_pos'])
    a, p = get_axis_from_name(axis).double(), get_axis_from_name(plane_axis).double()
    r = reference_frame_rotation_matrix(a_pos, p_pos, a, p, project_on_positive_axis=positive)
    np.testing.assert_allclose(r.numpy(), g[f'frame/{axis}{plane_axis}{int(positive)}'], atol=1e-12)
    # proper rotations that do what the contract says
    eye = torch.eye(3, dtype=torch.float64).expand(len(r), 3, 3)
    assert torch.allclose(torch.matmul(r, r.transpose(1, 2)), eye, atol=1e-12)
    assert torch.allclose(torch.linalg.det(r), torch.ones(len(r), dtype=torch.float64), atol=1e-12)
    pts = batchwise_rotate(torch.stack([a_pos, p_pos], dim=1), r)
    off_axis = pts[:, 0] - (pts[:, 0] @ a).unsqueeze(1) * a
    assert off_axis.abs().max() < 1e-12
    if positive:
        assert (pts[:, 0] @ a > 0).all()
    normal = torch.linalg.cross(a, p)
    assert (pts[:, 1] @ normal).abs().max() < 1e-12
    back = batchwise_rotate(pts, r, inverse=True)
    assert torch.allclose(back[:, 0], a_pos, atol=1e-12)


def test_frame_rotation_is_differentiable_and_handles_points_already_in_place():
    a, p = get_axis_from_name('x').double(), get_axis_from_name('y').double()
    a_pos = torch.tensor([[2.0, 0.0, 0.0], [-3.0, 0.0, 0.0], [0.3, -0.2, 0.9]], dtype=torch.float64)
    p_pos = torch.tensor([[1.0, 1.0, 0.0], [0.5, -2.0, 0.0], [0.1, 0.7, -0.4]], dtype=torch.float64, requires_grad=True)
    r = reference_frame_rotation_matrix(a_pos, p_pos, a, p)
    assert torch.allclose(r[0], torch.eye(3, dtype=torch.float64), atol=1e-14)
    assert torch.allclose(r[1], torch.eye(3, dtype=torch.float64), atol=1e-14)
    r.sum().backward()
    assert torch.isfinite(p_pos.grad).all()


def test_atom_flattened_helpers():
    assert atom_to_flattened_indices(np.array([0, 2])).tolist() == [0, 1, 2, 6, 7, 8]
    assert atom_to_flattened_indices(torch.tensor([0, 2]), space_dimension=2).tolist() == [0, 1, 4, 5]
    x = torch.arange(24.0).reshape(2, 12)
    assert flattened_to_atom(x).shape == (2, 4, 3)
    assert flattened_to_atom(x, 2).shape == (2, 6, 2)
    assert flattened_to_atom(x[0]).shape == (4, 3)
    assert torch.equal(atom_to_flattened(flattened_to_atom(x)), x)
    assert torch.equal(atom_to_flattened(flattened_to_atom(x[0])), x[0])


def test_oriented_fixed_indices_and_errors():
    inner = torch.nn.Identity()
    assert OrientedFlow(inner)._fixed_indices.tolist() == [1, 2, 5]                    # x axis, xy plane
    assert OrientedFlow(inner, plane_point_idx=0)._fixed_indices.tolist() == [4, 5, 2]  # axis point defaults to 1
    assert OrientedFlow(inner, axis_point_idx=2, plane_point_idx=0, axis='z', plane='yz'
                        )._fixed_indices.tolist() == [6, 7, 0]
    assert OrientedFlow(inner, axis_point_idx=3, plane_point_idx=1, axis='y', plane='xy'
                        )._fixed_indices.tolist() == [9, 11, 5]
    with pytest.raises(ValueError, match='must be different'):
        OrientedFlow(inner, axis_point_idx=1, plane_point_idx=1)
    with pytest.raises(ValueError, match='same plane'):
        OrientedFlow(inner, axis='z', plane='xy')
    with pytest.raises(ValueError, match='return_partial'):
        OrientedFlow(inner, return_partial=True)
    with pytest.raises(ValueError, match='inverse of OrientedFlow'):
        OrientedFlow(inner, rotate_back=False).inverse(torch.zeros(1, 9))


def test_centroid_fixed_indices_and_errors():
    inner = torch.nn.Identity()
    assert CenteredCentroidFlow(inner, space_dimension=3)._fixed_indices.tolist() == [0, 1, 2]
    assert CenteredCentroidFlow(inner, space_dimension=3, fixed_point_idx=1)._fixed_indices.tolist() == [3, 4, 5]
    f = CenteredCentroidFlow(inner, space_dimension=3, subset_point_indices=[1, 2], fixed_point_idx=1)
    assert f._fixed_indices.tolist() == [6, 7, 8]
    f = CenteredCentroidFlow(inner, space_dimension=2, subset_point_indices=[3, 1], weights=[1.0, 3.0])
    assert f._fixed_indices.tolist() == [6, 7]
    assert torch.allclose(f._weights, torch.tensor([[0.25], [0.75]]))
    assert f.space_dimension == 2
    with pytest.raises(ValueError, match='return_partial'):
        CenteredCentroidFlow(inner, space_dimension=3, return_partial=True)
    with pytest.raises(ValueError, match="'origin' must have length"):
        CenteredCentroidFlow(inner, space_dimension=3, origin=[0.0, 1.0])
    with pytest.raises(ValueError, match="'weights' must have the same length"):
        CenteredCentroidFlow(inner, space_dimension=3, subset_point_indices=[0, 1], weights=[1.0])
    with pytest.raises(ValueError, match='inverse of CenteredCentroidFlow'):
        CenteredCentroidFlow(inner, space_dimension=3, translate_back=False).inverse(torch.zeros(1, 9))


def test_wrapper_state_dict_keys_match_reference_fixture():
    g = gu.load('wrappers.npz')
    import tfep_amd.nn.flows as flows
    from tfep_amd.nn.flows import MAF
    from tfep_amd.nn.transformers import AffineTransformer, NeuralSplineTransformer
    from oracle.made import generate_degrees
    for name, cfg in gu.wrapper_configs().items():
        n_in = gu.wrapper_n_inner(cfg)
        if cfg.get('spline'):
            tr = NeuralSplineTransformer(x0=torch.full((n_in,), -8.0), xf=torch.full((n_in,), 8.0), n_bins=6)
        else:
            tr = AffineTransformer()
        inner = MAF(degrees_in=torch.as_tensor(generate_degrees(n_in, 'ascending')), transformer=tr,
                    initialize_identity=False)
        flow = gu.build_wrapped(cfg, inner, flows)
        gold = gu.sub(g, f'{name}/sd/')
        ours = {k: v for k, v in flow.state_dict().items() if not k.endswith('.mask')}
        assert set(ours) == set(gold), (name, set(ours) ^ set(gold))
        for k, v in ours.items():
            assert tuple(v.shape) == gold[k].shape, (name, k)
            if not v.is_floating_point() or k.split('.')[-1] in ('_axis', '_plane_axis', '_plane_normal', 'origin', '_weights'):
                np.testing.assert_allclose(v.numpy(), gold[k], rtol=1e-7, err_msg=f'{name} {k}')
