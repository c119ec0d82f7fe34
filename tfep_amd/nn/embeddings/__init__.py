from .mafembed import FlipInvariantEmbedding, MAFEmbedding, MixedEmbedding, PeriodicEmbedding  # noqa: F401
