"""Host side of the astts_op_* tensor operators (synthesis path).  Filled in as kernels land."""
from . import _lib

_SIGS = {}
_lib.register_signatures(_SIGS)
