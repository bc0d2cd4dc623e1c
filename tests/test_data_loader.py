"""load_csv_dataset (the counterpart of the reference driver's preprocessing, code/main.py:20-41) against what the reference's own
lines produce.  tests/golden/loader_main_py.npz was made by EXECUTING main.py:20-41 from the reference file (make_golden.py
`loader`) on the two data sets that script can select; it holds the raw CSV values and the resulting (XX, t)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from riemannhamiltonianmontecarlo_amd.data import load_csv_dataset


@pytest.mark.parametrize("name", ["australian", "heart"])
def test_loader_matches_reference_driver(tmp_path, name):
    g = np.load(os.path.join(GOLDEN, "loader_main_py.npz"))
    raw = g[name + "_raw"]
    path = tmp_path / (name + ".csv")
    np.savetxt(path, raw, delimiter=",", fmt="%.17g")                  # round-trips every double exactly
    assert np.array_equal(np.loadtxt(path, delimiter=","), raw)
    XX, t = load_csv_dataset(str(path))
    assert XX.shape == g[name + "_XX"].shape and t.shape == g[name + "_t"].shape == (raw.shape[0], 1)
    assert np.array_equal(t, g[name + "_t"])                            # heart: labels {1,2} -> {0,1} (main.py:26-27)
    assert set(np.unique(t)) == {0.0, 1.0}
    assert np.abs(XX - g[name + "_XX"]).max() <= 1e-15                  # same expressions up to the order of two roundings
    assert np.array_equal(XX[:, 0], np.ones(raw.shape[0]))              # intercept column first (main.py:40-41)
    # and the committed data fixtures the GPU tests use are exactly this
    d = np.load(os.path.join(GOLDEN, "data_%s.npz" % name))
    assert np.array_equal(d["XX"], XX) and np.array_equal(d["t"], t)


def test_german_label_remap_and_ripley_cubic_basis():
    """german has labels {1,2} like heart (remapped: main.py:26-27 / BLR_RMHMC.m:52-55); ripley uses the authors' cubic basis
    [1, X, X^2, X^3] of the standardised covariates, D = 7 (BLR_RMHMC.m:155-173)."""
    g = np.load(os.path.join(GOLDEN, "data_german.npz"))
    assert set(np.unique(g["t"])) == {0.0, 1.0} and g["XX"].shape == (1000, 25)
    r = np.load(os.path.join(GOLDEN, "data_ripley.npz"))["XX"]
    assert r.shape == (250, 7) and np.array_equal(r[:, 0], np.ones(250))
    assert np.allclose(r[:, 3:5], r[:, 1:3] ** 2) and np.allclose(r[:, 5:7], r[:, 1:3] ** 3)
    assert np.allclose(r[:, 1:3].mean(0), 0, atol=1e-12) and np.allclose(r[:, 1:3].std(0), 1)
