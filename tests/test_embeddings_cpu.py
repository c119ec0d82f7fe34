"""CPU tests of the flip-invariant / mixed embeddings: the numpy oracle against the reference's outputs, and the
host logic of the tfep_amd modules (degrees, errors, state_dict schema, flip invariance) -- these modules are
plain torch ops, so they also run on CPU tensors (reference tests/nn/embeddings/test_mafembed.py)."""
import numpy as np
import pytest
import torch

import golden_util as gu
import tfep_amd.nn.embeddings as E
from oracle import transformers as otr


def _nets(sd, prefix=''):
    def net(name):
        return tuple(sd[f'{prefix}{name}.{i}.{w}'].astype(np.float64) for i in (0, 2) for w in ('weight', 'bias'))
    return net('embedding_layer'), net('weight_layer')


def _oracle_embed(cfg, sd, x, prefix=''):
    if cfg['kind'] == 'flip':
        idx = sd[prefix + '_embedded_indices']
        non = sd[prefix + '_nonembedded_indices']
        e, w = _nets(sd, prefix)
        return otr.flip_invariant_embedding(x, idx, non, cfg.get('vector_dimension', 4), e, w)
    if cfg['kind'] == 'periodic':
        return otr.periodic_embedding(x, tuple(cfg['limits']), sd[prefix + '_periodic_indices'],
                                      sd[prefix + '_nonperiodic_indices'])
    parts = []
    for i, c in enumerate(cfg['layers']):
        p = f'{prefix}embedding_layers.{i}.'
        parts.append((sd[f'{prefix}_embedded_indices{i}'], lambda xs, c=c, p=p: _oracle_embed(c, sd, xs, p)))
    return otr.mixed_embedding(x, parts, sd[prefix + '_nonembedded_indices'])


@pytest.mark.parametrize('name', list(gu.embedding_configs()))
def test_oracle_embedding_matches_reference(name):
    g = gu.load('embeddings.npz')
    cfg = gu.embedding_configs()[name]
    sd = gu.sub(g, f'emb/{name}/sd/')
    out = _oracle_embed(cfg, sd, g[f'emb/{name}/x'].astype(np.float64))
    np.testing.assert_allclose(out, g[f'emb/{name}/out_f64'], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize('name', list(gu.embedding_configs()))
def test_module_matches_reference_schema_degrees_and_values(name):
    g = gu.load('embeddings.npz')
    cfg = gu.embedding_configs()[name]
    emb = gu.build_embedding(cfg, E)
    gold = gu.sub(g, f'emb/{name}/sd/')
    sd = emb.state_dict()
    assert set(sd) == set(gold), set(sd) ^ set(gold)
    for k in sd:
        assert tuple(sd[k].shape) == gold[k].shape and str(sd[k].dtype).split('.')[-1] == str(gold[k].dtype), k
    emb.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in gold.items()})
    deg = emb.get_degrees_out(torch.as_tensor(cfg['degrees_in']))
    assert deg.tolist() == g[f'emb/{name}/degrees_out'].tolist()
    if cfg['kind'] == 'flip':                       # pure torch ops: runs on CPU
        out = emb(torch.from_numpy(g[f'emb/{name}/x']))
        np.testing.assert_allclose(out.detach().numpy(), g[f'emb/{name}/out_f64'], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize('n_features_in,embedding_dimension,embedded_indices', [
    (4, 4, None), (12, 3, None), (9, 3, [2, 3, 4, 5]), (11, 2, [1, 2, 3, 4, 6, 7, 8, 9])])
def test_flip_invariance(n_features_in, embedding_dimension, embedded_indices):
    torch.manual_seed(0)
    emb = E.FlipInvariantEmbedding(n_features_in, embedding_dimension, embedded_indices)
    x = torch.randn(3, n_features_in)
    out, flipped = emb(x), emb(-x)
    n_vec = (n_features_in if embedded_indices is None else len(embedded_indices)) // emb.vector_dimension
    assert out.shape == (3, n_features_in + n_vec * (embedding_dimension - emb.vector_dimension))
    n_emb = n_vec * embedding_dimension
    assert torch.equal(out[:, -n_emb:], flipped[:, -n_emb:])
    assert torch.allclose(out[:, :-n_emb], -flipped[:, :-n_emb])


def test_embedding_errors():
    with pytest.raises(ValueError, match='Found duplicated indices'):
        E.FlipInvariantEmbedding(n_features_in=5, embedding_dimension=3, embedded_indices=[3, 3, 4])
    emb = E.FlipInvariantEmbedding(n_features_in=4, embedding_dimension=3)
    with pytest.raises(ValueError, match='same degree must be assigned'):
        emb.get_degrees_out(torch.tensor([0, 0, 0, 1]))
    with pytest.raises(ValueError, match='Different number of layers'):
        E.MixedEmbedding(4, [E.PeriodicEmbedding(1, [0., 1.])], [[0], [1]])
    with pytest.raises(ValueError, match='different feature indices'):
        E.MixedEmbedding(4, [E.PeriodicEmbedding(1, [0., 1.]), E.PeriodicEmbedding(2, [0., 1.])], [[0], [0, 1]])


@pytest.mark.parametrize('periodic_indices,flip_indices,degrees_in,expected', [
    ([0], [2, 3], [0, 2, 1, 1], [2, 0, 0, 1]), ([3], [0, 1], [2, 2, 0, 1], [0, 1, 1, 2]),
    ([2], [0, 1], [0, 0, 1, 2], [2, 1, 1, 0]), ([3], [1, 2], [1, 0, 0, 2], [1, 2, 2, 0])])
def test_mixed_embedding_get_degrees_out(periodic_indices, flip_indices, degrees_in, expected):
    """Cases of the reference's tests/nn/embeddings/test_mafembed.py:243-281."""
    mixed = E.MixedEmbedding(len(degrees_in), [
        E.PeriodicEmbedding(len(periodic_indices), limits=[0., 1.]),
        E.FlipInvariantEmbedding(len(flip_indices), embedding_dimension=1, vector_dimension=2)],
        [periodic_indices, flip_indices])
    assert mixed.get_degrees_out(torch.tensor(degrees_in)).tolist() == expected
