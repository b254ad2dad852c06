"""CPU: the C-ABI library builds/loads and exports every symbol include/nmpc_solver.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from mpc_trajectory_generator_amd import _lib, named_config
from mpc_trajectory_generator_amd.solver import problem_from_config


def test_header_symbols_exported():
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "nmpc_solver.h")).read()
    declared = set(re.findall(r"\b(nmpc_[a-z0-9_]+)\s*\(", header))
    declared -= {"nmpc_handle"}
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.nmpc_abi_version() == _lib.EXPECTED_ABI == 3


def test_sizes_and_struct_layout():
    lib = _lib.load_library()
    for name, (nu, np_, n1, n2) in {"cfg1": (40, 430, 40, 13), "cfg2": (80, 810, 80, 13),
                                    "cfg3": (40, 550, 40, 53), "cfg4": (40, 430, 40, 13)}.items():
        cfg = named_config(name)
        pb = problem_from_config(cfg)
        assert (cfg.n_u, cfg.n_p, cfg.n1, cfg.n2) == (nu, np_, n1, n2)
        assert lib.nmpc_n_u(ctypes.byref(pb)) == nu and lib.nmpc_n_p(ctypes.byref(pb)) == np_
        assert lib.nmpc_n1(ctypes.byref(pb)) == n1 and lib.nmpc_n2(ctypes.byref(pb)) == n2
    assert ctypes.sizeof(_lib.NmpcProblem) == 72 and ctypes.sizeof(_lib.NmpcOpts) == 88
    o = _lib.NmpcOpts()
    lib.nmpc_default_opts(ctypes.byref(o))
    assert (o.tolerance, o.lbfgs_memory, o.max_inner, o.max_outer, o.initial_penalty) == (1e-4, 10, 500, 10, 1.0)
    # the budget is off; akkt_gradient defaults to 1 (step_top), the other switches to 0
    assert (o.max_total_inner, o.akkt_gradient, o.ls_failure, o.inner_status) == (0, 1, 0, 0)


def test_oracle_and_abi_share_option_fields():
    """The test oracle's option struct mirrors nmpc_opts field for field (the parity tests pass one dict to both)."""
    from oracle.binding import OrcOpts
    assert [f for f, _ in OrcOpts._fields_] == [f for f, _ in _lib.NmpcOpts._fields_]
    assert ctypes.sizeof(OrcOpts) == ctypes.sizeof(_lib.NmpcOpts)


def test_no_device_fails_loudly():
    """Without a GPU the product must raise, never fall back to a CPU path."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from mpc_trajectory_generator_amd.solver import BatchSolver, SolverError
    with pytest.raises(SolverError):
        BatchSolver(named_config("default"), max_batch=4)


def test_shipped_library_reads_no_environment():
    """The shipped .so carries no NMPC_* knob (the interface it replaces has none, src/path_generator.py:218-222): no such string in
    the binary, and it says so itself.  Knobs exist only in the experiments variant the tests and scripts build."""
    path = _lib.build_library()
    blob = open(path, "rb").read()
    assert blob.count(b"NMPC_") == 0, sorted(set(re.findall(rb"NMPC_[A-Z0-9_]+", blob)))
    assert b"getenv" not in blob
    lib = _lib.load_library()
    assert lib.nmpc_experiments_build() == 0


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "mpc_trajectory_generator_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle\b", text, re.M), f
                assert not re.search(r"#\s*include[^\n]*oracle", text), f
                assert "libnmpc_oracle" not in text and "dlopen" not in text, f
