"""CPU oracle (test infrastructure, NOT the product) -- see oracle/nmpc_oracle.h.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from .binding import Oracle, OrcProblem, OrcOpts, OrcStatus, build_oracle  # noqa: F401
