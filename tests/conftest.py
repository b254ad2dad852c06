import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


GOLDEN = os.path.join(ROOT, "tests", "golden")
VARIANTS = {"default": "cfg1", "n40": "cfg2", "nobs50": "cfg3", "smooth": "cfg4"}


def oracle_for(cfg, **opts):
    from oracle import Oracle
    return Oracle(cfg.N_hor, cfg.Nobs, cfg.Ndynobs, cfg.ts, cfg.lin_vel_min, cfg.lin_vel_max,
                  cfg.ang_vel_max, cfg.lin_acc_min, cfg.lin_acc_max, cfg.ang_acc_max, **opts)


@pytest.fixture(scope="session")
def golden():
    return {k: np.load(os.path.join(GOLDEN, f"cost_{k}.npz")) for k in VARIANTS}


STATUS_FIELDS = ("exit_status", "num_outer_iterations", "num_inner_iterations", "num_cost_evals",
                 "num_grad_evals", "last_problem_norm_fpr", "delta_y_norm_over_c", "f2_norm", "penalty", "cost")
