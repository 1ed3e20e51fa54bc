"""rts_amd -- MI355X-native hot path of the RTS ray-traced radar return simulator.

The product is the C-ABI shared library rts_amd/librts_amd.so (sources in rts_amd/csrc,
interface in include/rts_amd.h, C++ adapter in include/rts_adapter.hpp).  This Python package
is only the harness that tests and bench.py use to drive that library through ctypes.
"""
from . import _lib  # noqa: F401
from ._lib import PRD_DTYPE, GROUP_DTYPE, RESPONSE_DTYPE, RtsError, build  # noqa: F401
