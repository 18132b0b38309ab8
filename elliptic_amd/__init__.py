"""elliptic_amd -- MI355X-native batched elliptic-curve scalar multiplication
behind indutny/elliptic's Point.mul / mulAdd / jmulAdd / ec.verify hot path.

The product is libellgpu.so (hand-written HIP for gfx950 + a C ABI,
include/ellgpu.h); this package is the thin Python host layer used by the
tests and bench.py.  The JS host layer the reference's users switch to lives in
elliptic_amd/js (N-API addon + install() patch).
"""
from .engine import (CURVES, CURVE_ID, FIELD_BYTES, ORDER_BYTES, STATUS_OFF_CURVE, Context, be_to_ints,  # noqa: F401
                     ints_to_be)
from ._lib import EllgpuError  # noqa: F401

__version__ = "0.1.0"
