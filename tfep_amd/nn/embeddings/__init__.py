from .mafembed import FlipInvariantEmbedding, MAFEmbedding, MixedEmbedding, PeriodicEmbedding  # noqa: F401
from .radial import (BehlerParrinelloRadialExpansion, GaussianBasisExpansion,  # noqa: F401
                     behler_parrinello_cosine_switching_function)
