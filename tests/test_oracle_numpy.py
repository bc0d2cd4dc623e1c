"""Pin the NumPy restatement of the reference's literal algorithm (oracle/rmhmc_numpy.py: the stand-in for the "reference NumPy path"
that bench.py's cpu_baseline times on the GPU box) to the golden tapes captured from code/rmhmc.py.  CPU only."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, load_tape, rel_err

sys.path.insert(0, os.path.join(ROOT, "oracle"))
import rmhmc_numpy as rn  # noqa: E402


@pytest.mark.parametrize("name", ["pima", "heart", "ripley", "syn_m1000_d8", "syn_m50_d5_L1", "syn_m300_d20", "syn_m203_d33", "guard_w"])
def test_numpy_restatement_replays_the_reference(name):
    XX, t, g = load_tape(name)
    t = t.ravel()
    T = min(8, g["z"].shape[0])
    checked = 0
    for it in range(T):
        u_acc = 0.5 if np.isnan(g["u_acc"][it]) else float(g["u_acc"][it])
        r = rn.transition(XX, t, g["w_before"][it], g["z"][it], float(g["u_len"][it]), float(g["g_dir"][it]), u_acc, L=int(g["L"]),
                          eps=float(g["eps"]), K=int(g["K"]))
        assert r["nsteps"] == int(g["nsteps"][it])
        if not np.isfinite(g["H_prop"][it]):
            assert not r["accepted"]
            continue
        assert rel_err(r["w_prop"], g["w_prop"][it]) < 1e-8 and rel_err(r["p_prop"], g["p_prop"][it]) < 1e-8, it
        assert abs(r["hld_prop"] - g["hld_prop"][it]) < 1e-8 * max(1.0, abs(g["hld_prop"][it]))
        assert abs(r["H_prop"] - g["H_prop"][it]) < 1e-7 * max(1.0, abs(g["H_prop"][it]))
        assert abs(r["H_cur"] - g["H_cur"][it]) < 1e-9 * max(1.0, abs(g["H_cur"][it]))
        assert rel_err(r["w"], g["w_after"][it]) < 1e-8
        checked += 1
    assert checked >= T // 2
