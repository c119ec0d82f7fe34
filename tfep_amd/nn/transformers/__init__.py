from .transformer import Transformer, MAFTransformer  # noqa: F401
from .affine import AffineTransformer, VolumePreservingShiftTransformer  # noqa: F401
from .spline import NeuralSplineTransformer  # noqa: F401
from .moebius import MoebiusTransformer  # noqa: F401
from .mixed import MixedTransformer  # noqa: F401
