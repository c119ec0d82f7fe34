from .mafembed import MAFEmbedding, PeriodicEmbedding  # noqa: F401
