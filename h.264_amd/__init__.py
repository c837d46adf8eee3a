"""h.264_amd -- MI355X (gfx950) implementation of the JM lencod per-macroblock hot path.

The product is the C-ABI shared library `libjmhip.so` (include/jmhip.h, hand-written HIP under csrc/).
This package is the thin Python binding over that ABI used by the tests and bench.py. It never computes
anything itself and has no CPU fallback: if the library is missing or no GPU is present the calls raise.

The directory name contains a dot, so load it by path (tests/conftest.py::load_pkg, __graft_entry__.py).
"""
from .jmhip import (  # noqa: F401
    JmhipError, Context, MeParams, BipredParams, DeblockParams, SliceParams, MB_INTER_DTYPE, MB_BIPRED_DTYPE, SLICE_REFS, load_library, library_path, declared_symbols, partition_table,
    build_library, flat_quant, NPART, PAD, STAGES,
    ME_MB_DTYPE, ME_RESULT_DTYPE, QUANT_DTYPE, TQ_JOB_DTYPE, TQ_RESULT_DTYPE, DIST_JOB_DTYPE, MB_MODE_DTYPE,
    SURFACE_JOB_DTYPE, BIPRED_JOB_DTYPE, BIPRED_RESULT_DTYPE, PREDCOST_JOB_DTYPE, DEBLOCK_MB_DTYPE, DEBLOCK_BLK_DTYPE,
)
from . import slices  # noqa: F401,E402
from . import slice_host  # noqa: F401,E402
