import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before the HIP library is dlopen-ed: one HIP runtime per process, see _capi.load_hip_library)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
ORACLE_LIB = os.path.join(ROOT, "oracle", "librmhmc_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


from riemannhamiltonianmontecarlo_amd import _capi  # noqa: E402
from riemannhamiltonianmontecarlo_amd.data import synthetic_logreg  # noqa: E402


def _build_oracle():
    import subprocess
    src = os.path.join(ROOT, "oracle", "rmhmc_oracle.c")
    if (not os.path.exists(ORACLE_LIB)) or os.path.getmtime(ORACLE_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle bound through the same ctypes class as the product (test infrastructure)."""
    _build_oracle()
    return _capi.RmhmcLib(ORACLE_LIB)


@pytest.fixture(scope="session")
def hip():
    """The product library; GPU tests fail loudly if it is not built."""
    return _capi.load_hip_library()


# (the last two: the blocked large-D path, 64 < D <= 256 - three column blocks with a ragged M, and BASELINE config 5's own shape, whose
#  tape holds vectors, scalars and matrix DIAGONALS only: tests/golden/make_golden.py large_d_tapes)
TAPES = ["pima", "australian", "german", "heart", "ripley", "syn_m1000_d8", "syn_m50_d5_L1", "syn_m300_d20", "syn_m203_d33",
         "syn_m10000_d64_L1", "guard_w", "syn_m3001_d130", "syn_m50000_d256_L1"]
LITERAL_TOO_SLOW = ("syn_m10000_d64_L1", "syn_m3001_d130", "syn_m50000_d256_L1")  # O(M D^3) tensor in scalar C: minutes


def load_tape(name):
    """Returns (XX, t, tape dict) for a golden transition tape captured from the reference."""
    g = dict(np.load(os.path.join(GOLDEN, "tape_%s.npz" % name)))
    if name in ("pima", "australian", "german", "heart", "ripley"):   # (ripley: the authors' cubic basis, D = 7)
        d = np.load(os.path.join(GOLDEN, "data_%s.npz" % name))
        XX, t = d["XX"], d["t"]
    else:
        XX, t = synthetic_logreg(int(g["M"]), int(g["D"]), int(g["data_seed"]))
        if "x_scale" in g:
            XX = XX * float(g["x_scale"])
    return XX, t, g


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def mat_err(G, g, key):
    """rel_err of a D x D matrix against the tape's copy, or of its diagonal when the tape is a compact one (``diag`` + key)."""
    for cut in range(len(key), 0, -1):                       # "it0_s0_G_end" -> "it0_s0_" + "diag" + "G_end"
        if key[:cut].endswith("_") and (key[:cut] + "diag" + key[cut:]) in g:
            return rel_err(np.diag(np.asarray(G)), g[key[:cut] + "diag" + key[cut:]])
    return rel_err(G, g[key])


def logdet_after_first_step(g):
    """log|G| at the end of the first leapfrog step of transition 0, from the reference's own values: slogdet of its G where the tape
    holds the matrix; for a compact tape (one-step trajectories) 2 x ProposedLogDet (rmhmc.py:171)."""
    if "it0_s0_G_end" in g:
        sign, ld = np.linalg.slogdet(g["it0_s0_G_end"])
        assert sign > 0
        return float(ld)
    assert int(g["nsteps"][0]) == 1
    return 2.0 * float(g["hld_prop"][0])
