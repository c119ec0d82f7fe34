from .conditioner import Conditioner  # noqa: F401
from .made import MADE, generate_degrees  # noqa: F401
