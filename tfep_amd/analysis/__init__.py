from .estimator import fep_estimator  # noqa: F401
from .bootstrap import bootstrap, bootstrap_fep  # noqa: F401
