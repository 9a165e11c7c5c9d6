"""csolve_amd -- MI355X (gfx950) implementation of CSolve's constraint-propagation
fixpoint behind a C ABI (include/csolve_gpu.h, libcsolve_hip.so).

Python is glue only: it hands torch device pointers and streams to the C ABI.  There is
no CPU implementation of propagation in this package; importing `csolve_amd.solver`
without a built libcsolve_hip.so raises, and calling into it without a HIP device fails
with the library's error.
"""
from ._lib import LIB_PATH, CsolveError, load_library, declared_symbols  # noqa: F401
