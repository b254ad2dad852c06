"""MI355X-native batched NMPC trajectory solver (hot path of wljungbergh/mpc-trajectory-generator).

Host side (this package, Python) mirrors the reference's solver-facing interface; the solve itself
runs in hand-written HIP kernels behind the C-ABI declared in include/nmpc_solver.h.
"""
from .config import Config, load_config, named_config  # noqa: F401
