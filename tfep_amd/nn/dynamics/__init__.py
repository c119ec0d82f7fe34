from .egnn import EGNNDynamics  # noqa: F401
