"""Masked autoregressive flow (reference ``tfep/nn/flows/maf.py:33-194``)."""
from typing import Optional, Sequence, Union

import torch

from ...utils.misc import ensure_tensor_sequence
from ..conditioners.made import MADE
from ..embeddings.mafembed import MAFEmbedding
from ..transformers.affine import AffineTransformer
from .autoregressive import AutoregressiveFlow


class MAF(AutoregressiveFlow):
    """Autoregressive flow with a MADE conditioner and an arbitrary transformer.

    >>> from tfep_amd.nn.conditioners.made import generate_degrees
    >>> maf = MAF(degrees_in=generate_degrees(n_features=5, order='ascending'))
    >>> maf = MAF(degrees_in=[-1, -1, -1, 0, 0, 1, 2])   # first 3 features are conditioning

    Arguments as reference maf.py:82-129.
    """

    def __init__(
            self,
            degrees_in: Sequence[int],
            transformer: Optional[torch.nn.Module] = None,
            hidden_layers: Union[int, Sequence[int], Sequence[Sequence[int]]] = 2,
            embedding: Optional[MAFEmbedding] = None,
            weight_norm: bool = True,
            initialize_identity: bool = True,
    ):
        degrees_in = ensure_tensor_sequence(degrees_in)
        highest = _check_degrees(degrees_in)
        transformer = AffineTransformer() if transformer is None else transformer
        # degree d >= 0: the features transformed at step d of the inverse; -1: conditioning features
        groups = [torch.nonzero(degrees_in == d).flatten() for d in range(highest + 1)]
        conditioner = _EmbeddedMADE(
            embedding=embedding,
            degrees_in=degrees_in if embedding is None else embedding.get_degrees_out(degrees_in),
            degrees_out=transformer.get_degrees_out(degrees_in[degrees_in != -1]),
            hidden_layers=hidden_layers,
            weight_norm=weight_norm,
        )
        super().__init__(n_features_in=len(degrees_in), transformer_indices=groups, conditioner=conditioner,
                         transformer=transformer, initialize_identity=initialize_identity)
        self._embedding = embedding

    def n_parameters(self) -> int:
        """The total number of (unmasked) parameters."""
        return self._conditioner.n_parameters()


def _check_degrees(degrees_in):
    """Degrees must be -1 (conditioning) or 0, 1, 2, ... without gaps; returns the largest one."""
    present = sorted(set(degrees_in.tolist()))
    lowest, highest = present[0], present[-1]
    if lowest not in (-1, 0) or present != list(range(lowest, highest + 1)):
        raise ValueError('degrees_in must assume consecutive values starting '
                         'from 0 (or -1 for conditioning input features).')
    return highest


class _EmbeddedMADE(MADE):
    """A MADE conditioner with embedded input features (reference maf.py:184-194)."""

    def __init__(self, embedding, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.embedding = embedding

    def _embed(self, x):
        return x if self.embedding is None else self.embedding(x)
