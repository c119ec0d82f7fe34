"""Log formats of the reference's ``tfep.io`` that the hot path writes to (``tfep/io/log.py``)."""
from .log import TFEPLogger, gather_to_rank0

__all__ = ['TFEPLogger', 'gather_to_rank0']
