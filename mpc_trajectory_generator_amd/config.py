"""Configuration for the batched NMPC solver: same YAML keys as the reference.

Mirrors reference src/utils/config.py:8-72 (34 required keys, a missing key raises) and carries
the values of reference configs/default.yaml as built-in defaults so that no reference file has
to travel with this package.  Keys under "essential for the mpc-formulation"
(configs/default.yaml:6-13) plus the size keys (:34-40) and ts (:18) are what the reference bakes
into the generated solver; here they select the kernel instantiation and fill ``nmpc_problem``.
"""
from __future__ import annotations

import yaml

# reference configs/default.yaml:7-50
DEFAULTS = dict(
    N_hor=20, lin_vel_min=-0.5, lin_vel_max=1.5, lin_acc_min=-1, lin_acc_max=1,
    ang_vel_max=0.5, ang_acc_max=3,
    throttle_ratio=1.0, num_steps_taken=1, ts=0.2, vel_red_steps=20,
    lin_vel_penalty=0, lin_acc_penalty=10.0, ang_vel_penalty=0, ang_acc_penalty=5.0,
    cte_penalty=200, q=0.0, qv=10.0, qtheta=0.0, qN=0.0, qthetaN=0.0,
    nx=3, nz=20, nu=2, nobs=3, Nobs=10, Ndynobs=3, ndynobs=5,
    vehicle_width=0.5, vehicle_margin=0.25,
    build_type="release", build_directory="mpc_build",
    bad_exit_codes=["NotConvergedIterations", "NotConvergedOutOfTime"],
    optimizer_name="navigation",
)

REQUIRED = tuple(DEFAULTS.keys())          # reference src/utils/config.py:8-43

# reference configs/smooth_velocity.yaml as committed lacks qv / vel_red_steps / Ndynobs and has
# N_hor 15, nz 19 (the reference's own Configurator rejects it); BASELINE.md section 4 row 4
# defines the overlay used here: its distinctive values on top of default's missing keys.
SMOOTH_VELOCITY = dict(
    ang_vel_max=1, ang_acc_max=5, throttle_ratio=0.9, num_steps_taken=2,
    lin_acc_penalty=8.0, ang_acc_penalty=20.0, cte_penalty=20.0, q=1, qtheta=0, qN=5, qthetaN=0.2,
    build_type="debug",
)
# reference configs/jconf_3.yaml:23 differs from default only here; BASELINE config 2 adds N_hor=40
JCONF_3 = dict(lin_acc_penalty=100.0)


class Config(dict):
    """dot-access dict, like the reference's ``dotdict`` (src/utils/config.py:46-50), except that
    an unknown attribute raises instead of silently returning None."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    # sizes the reference derives in mpc_generator.py:70-71,79
    @property
    def n_u(self):
        return self.nu * self.N_hor

    @property
    def n_p(self):
        return (self.nz + self.N_hor + self.Nobs * self.nobs
                + self.Ndynobs * self.ndynobs * self.N_hor + self.nx * self.N_hor)

    @property
    def n1(self):
        return 2 * self.N_hor

    @property
    def n2(self):
        return self.Nobs + self.Ndynobs

    def weights(self):
        """The ten run-time weights in the order the reference sends them
        (src/path_generator.py:226-227 -> p[10:20])."""
        return [float(self.q), float(self.qv), float(self.qtheta), float(self.lin_vel_penalty),
                float(self.ang_vel_penalty), float(self.qN), float(self.qthetaN),
                float(self.cte_penalty), float(self.lin_acc_penalty), float(self.ang_acc_penalty)]


def load_config(yaml_path: str | None = None, **overrides) -> Config:
    """YAML (optional) + overrides -> Config.  With a YAML path every required key must be
    present in the file (reference src/utils/config.py:60-65); without one the built-in
    default.yaml values are used."""
    if yaml_path is None:
        cfg = Config(DEFAULTS)
    else:
        with open(yaml_path, "r") as fh:
            raw = yaml.safe_load(fh)
        cfg = Config()
        for key in REQUIRED:
            if raw.get(key) is None and key not in overrides:
                raise RuntimeError(f"[CONFIG] Configuration is not properly set: missing '{key}'")
            cfg[key] = raw.get(key)
    cfg.update(overrides)
    if (cfg.nz, cfg.nu, cfg.nx, cfg.nobs, cfg.ndynobs) != (20, 2, 3, 3, 5):
        # mpc_generator.py:73-75 unpacks z0[0..19]; the layout constants are not free parameters
        raise RuntimeError("[CONFIG] nz/nu/nx/nobs/ndynobs must be 20/2/3/3/5")
    if not 1 <= cfg.num_steps_taken <= cfg.N_hor:
        raise RuntimeError("[CONFIG] num_steps_taken out of range")
    return cfg


def named_config(name: str) -> Config:
    """The five BASELINE.json configurations by name."""
    if name in ("default", "cfg0", "cfg1"):
        return load_config()
    if name in ("jconf_3_n40", "cfg2"):
        return load_config(**JCONF_3, N_hor=40)
    if name in ("nobs50", "cfg3"):
        return load_config(Nobs=50)
    if name in ("smooth_velocity", "cfg4"):
        return load_config(**SMOOTH_VELOCITY)
    raise KeyError(name)
