from .autoregressive import AutoregressiveFlow  # noqa: F401
from .maf import MAF  # noqa: F401
from .sequential import SequentialFlow  # noqa: F401
from .partial import PartialFlow  # noqa: F401
from .centroid import CenteredCentroidFlow  # noqa: F401
from .oriented import OrientedFlow  # noqa: F401
from .continuous import ContinuousFlow  # noqa: F401
from .pca import PCAWhitenedFlow  # noqa: F401
