"""bensolve_amd -- MI355X-native engine for the two inner loops of BENSOLVE's Benson algorithm.

The product is the C-ABI shared library ``bensolve_amd/csrc/libbslv_hip.so`` (include/bslv_hip.h).
This package is the Python-side mirror of the reference's host interface for that path
(bslv_lp.h / bslv_poly.h / bslv_algs.c's phase-2 loop), used by the tests and bench.py.
There is no CPU fallback: importing the engine classes without the built library raises.
"""
from ._lib import load_library, LibraryMissing  # noqa: F401
