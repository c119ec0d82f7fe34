from .estimator import fep_estimator  # noqa: F401
