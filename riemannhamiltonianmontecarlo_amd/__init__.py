"""MI355X-native RMHMC hot path (Bayesian logistic regression).

Host side of the C-ABI in include/rmhmc.h; mirrors the reference's
``code/rmhmc.py`` interface.  See DESIGN.md.
"""
from .rmhmc import RMHMC  # noqa: F401
from .hmc import HMC  # noqa: F401
from .mmala import mMALA  # noqa: F401
from . import tools, data, experiment  # noqa: F401

__all__ = ["RMHMC", "HMC", "mMALA", "tools", "data", "experiment"]
