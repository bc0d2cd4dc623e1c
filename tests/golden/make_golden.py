#!/usr/bin/env python3
"""Capture golden vectors from the reference implementation.

Runs ONLY in the build container (it imports /root/reference/code/rmhmc.py and
tools.py, which never travel to the GPU box).  It executes the reference's
``RMHMC`` under ``sys.settrace`` and records the values of its local variables
at fixed source lines, together with every random number the reference drew, so
that the oracle (oracle/rmhmc_oracle.c) and the HIP library can replay exactly
the same transitions.  Output: small ``.npz`` files next to this script.

    python tests/golden/make_golden.py            # regenerate everything
    python tests/golden/make_golden.py ripley loader ess4096     # only the named groups (base, ripley, loader, ess4096, larged)

The fixtures hold data only (inputs, random draws, expected outputs); no
reference source text is stored.
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = "/root/reference/code"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import rmhmc as ref_rmhmc  # noqa: E402  (the reference)
import tools as ref_tools  # noqa: E402

from riemannhamiltonianmontecarlo_amd.data import load_csv_dataset, synthetic_logreg  # noqa: E402

# ---------------------------------------------------------------------------
# Source lines of code/rmhmc.py at which locals are sampled.  Checked against
# the file text so that a changed reference fails loudly instead of silently
# capturing the wrong thing.
# ---------------------------------------------------------------------------
LINES = {
    96: "for StepNum in range(RandomStep)",
    102: "PM = ProposedMomentum.copy()",
    103: "for FixedIter in range(NumOfNewtonSteps)",
    110: "ProposedMomentum = PM",
    115: "for FixedIter in range(NumOfNewtonSteps)",
    123: "wNew = Pw",
    166: "LogPrior      = LogNormPDF",
    179: "Ratio = -ProposedH + CurrentH",
    190: "if IterationNum > BurnIn",
}


def _check_lines():
    src = open(os.path.join(REF, "rmhmc.py")).read().splitlines()
    for ln, frag in LINES.items():
        assert frag in src[ln - 1], (ln, src[ln - 1])


class Recorder:
    """Per-transition record of the reference's state."""

    def __init__(self, detail_iters, compact=False):
        self.detail_iters = detail_iters
        # compact: D x D matrices are reduced to their diagonals when recorded (the large-D tapes: five 512 KB matrices
        # per step would be most of the fixture directory; vectors, scalars and diag(G) pin the same code paths)
        self.mat = (lambda A: np.diag(A).copy()) if compact else (lambda A: A.copy())
        self.iters = []  # one dict per IterationNum
        self.cur = None
        self.draws = []  # (kind, value) in call order

    def tracer(self, frame, event, arg):
        if frame.f_code.co_name != "RMHMC":
            return None
        return self.local

    def local(self, frame, event, arg):
        if event != "line":
            return self.local
        ln = frame.f_lineno
        if ln not in LINES:
            return self.local
        L = frame.f_locals
        it = L["IterationNum"]
        if self.cur is None or self.cur["it"] != it:
            self.cur = {"it": it, "steps": [], "seen96": 0}
            self.iters.append(self.cur)
        c = self.cur
        detail = it < self.detail_iters
        if ln == 96:
            if c["seen96"] == 0:
                c["w"] = L["w"].copy().ravel()
                c["p0"] = L["ProposedMomentum"].copy().ravel()
                c["nsteps"] = int(L["RandomStep"])
                c["dir"] = int(L["TimeStep"])
                if detail:
                    c["G0"] = self.mat(L["G"])
                    c["InvG0"] = self.mat(L["InvG"])
                    c["tr0"] = L["TraceInvGdG"].copy().ravel()
                    c["cholG0"] = self.mat(L["OriginalCholG"])
            elif detail:
                st = c["steps"][-1]
                st["w_end"] = L["wNew"].copy().ravel()
                st["p_end"] = L["ProposedMomentum"].copy().ravel()
                st["G_end"] = self.mat(L["G"])
                st["tr_end"] = L["TraceInvGdG"].copy().ravel()
                st["grad_end"] = L["likelihood_grad"].copy().ravel()
            c["seen96"] += 1
        elif ln == 102 and detail:
            c["steps"].append({"grad_start": L["likelihood_grad"].copy().ravel(), "PM": [], "Pw": []})
        elif ln == 103 and detail:
            c["steps"][-1]["PM"].append(L["PM"].copy().ravel())
        elif ln == 115 and detail:
            c["steps"][-1]["Pw"].append(L["Pw"].copy().ravel())
        elif ln == 166:
            c["w_prop"] = L["wNew"].copy().ravel()
            c["p_prop"] = L["ProposedMomentum"].copy().ravel()
            if detail:
                c["G_prop"] = self.mat(L["G"])
                if c["steps"]:
                    st = c["steps"][-1]
                    st["w_end"] = c["w_prop"]
                    st["p_end"] = c["p_prop"]
                    st["G_end"] = self.mat(L["G"])
                    st["tr_end"] = L["TraceInvGdG"].copy().ravel()
                    st["grad_end"] = L["likelihood_grad"].copy().ravel()
        elif ln == 179:
            c["ljl_prop"] = float(np.ravel(L["ProposedLJL"])[0])
            c["hld_prop"] = float(L["ProposedLogDet"])
            c["H_prop"] = float(np.ravel(L["ProposedH"])[0])
            c["hld_cur"] = float(L["CurrentLogDet"])
            c["H_cur"] = float(np.ravel(L["CurrentH"])[0])
            c["ljl_cur"] = float(np.ravel(L["CurrentLJL"])[0])
        elif ln == 190:
            c["w_after"] = L["w"].copy().ravel()
            c["ratio"] = float(np.ravel(L["Ratio"])[0])
        return self.local


@contextlib.contextmanager
def recording_rng(rec):
    o_randn, o_rand = np.random.randn, np.random.rand

    def randn(*a):
        v = o_randn(*a)
        rec.draws.append(("randn", np.array(v, dtype=np.float64).ravel()))
        return v

    def rand(*a):
        v = o_rand(*a)
        rec.draws.append(("rand", np.array(v, dtype=np.float64).ravel()))
        return v

    np.random.randn, np.random.rand = randn, rand
    try:
        yield
    finally:
        np.random.randn, np.random.rand = o_randn, o_rand


def capture(XX, t, seed, n_iter, L=6, eps=0.5, K=4, detail_iters=2, compact=False):
    """Run the reference for n_iter transitions and return a flat dict of arrays.  compact=True stores diag(G) etc. under
    the keys ``*diagG0`` / ``*diagG_end`` ... instead of the full matrices."""
    rec = Recorder(detail_iters, compact)
    np.random.seed(seed)
    buf = io.StringIO()
    with recording_rng(rec), contextlib.redirect_stdout(buf), np.errstate(all="ignore"):
        sys.settrace(rec.tracer)
        try:
            # BurnIn = n_iter-1 < NumOfIterations as the reference requires (`start` is bound at
            # IterationNum == BurnIn, rmhmc.py:194-196)
            ref_rmhmc.RMHMC(XX, t, NumOfIterations=n_iter, BurnIn=n_iter - 1, NumOfLeapFrogSteps=L, StepSize=eps,
                            NumOfNewtonSteps=K)
        finally:
            sys.settrace(None)
    printed = buf.getvalue()
    T, D = n_iter, XX.shape[1]
    assert len(rec.iters) == T, (len(rec.iters), T)
    # split the draw log per transition: randn(1,D), rand(), randn(), [rand()]
    z = np.zeros((T, D)); u_len = np.zeros(T); g_dir = np.zeros(T); u_acc = np.full(T, np.nan)
    i = 0
    for it in range(T):
        k, v = rec.draws[i]; assert k == "randn" and v.size == D; z[it] = v; i += 1
        k, v = rec.draws[i]; assert k == "rand" and v.size == 1; u_len[it] = v[0]; i += 1
        k, v = rec.draws[i]; assert k == "randn" and v.size == 1; g_dir[it] = v[0]; i += 1
        if i < len(rec.draws) and rec.draws[i][0] == "rand" and rec.draws[i][1].size == 1 and (
                i + 1 >= len(rec.draws) or rec.draws[i + 1][1].size == D):
            # a second rand() before the next randn(1,D): the accept draw (rmhmc.py:181)
            if not rec.iters[it]["ratio"] > 0:
                u_acc[it] = rec.draws[i][1][0]; i += 1
    assert i == len(rec.draws), (i, len(rec.draws))
    out = {
        "seed": np.int64(seed), "L": np.int64(L), "eps": np.float64(eps), "K": np.int64(K),
        "z": z, "u_len": u_len, "g_dir": g_dir, "u_acc": u_acc,
        "w_before": np.stack([c["w"] for c in rec.iters]),
        "p0": np.stack([c["p0"] for c in rec.iters]),
        "nsteps": np.array([c["nsteps"] for c in rec.iters], dtype=np.int64),
        "dir": np.array([c["dir"] for c in rec.iters], dtype=np.int64),
        "w_prop": np.stack([c["w_prop"] for c in rec.iters]),
        "p_prop": np.stack([c["p_prop"] for c in rec.iters]),
        "w_after": np.stack([c["w_after"] for c in rec.iters]),
        "ratio": np.array([c["ratio"] for c in rec.iters]),
        "H_prop": np.array([c["H_prop"] for c in rec.iters]),
        "H_cur": np.array([c["H_cur"] for c in rec.iters]),
        "hld_prop": np.array([c["hld_prop"] for c in rec.iters]),
        "hld_cur": np.array([c["hld_cur"] for c in rec.iters]),
        "ljl_prop": np.array([c["ljl_prop"] for c in rec.iters]),
        "ljl_cur": np.array([c["ljl_cur"] for c in rec.iters]),
        "guard_p_fired": np.int64(printed.count("RENORMALIZE - ProposedMomentum")),
        "guard_w_fired": np.int64(printed.count("RENORMALIZE - wNew")),
    }
    for it in range(min(detail_iters, T)):
        c = rec.iters[it]
        pre = "it%d_" % it
        m = "diag" if compact else ""
        out[pre + m + "G0"] = c["G0"]; out[pre + m + "InvG0"] = c["InvG0"]; out[pre + "tr0"] = c["tr0"]
        out[pre + m + "cholG0"] = c["cholG0"]
        out[pre + m + "G_prop"] = c["G_prop"]
        assert len(c["steps"]) == c["nsteps"]
        for s, st in enumerate(c["steps"]):
            sp = pre + "s%d_" % s
            PM = st["PM"]; Pw = st["Pw"]
            assert len(PM) == K + 1 and len(Pw) == K + 1, (len(PM), len(Pw))
            out[sp + "grad_start"] = st["grad_start"]
            out[sp + "PM"] = np.stack(PM)   # PM[0] = p at step start, PM[k] after k fixed-point iterations
            out[sp + "Pw"] = np.stack(Pw)   # Pw[0] = w at step start, Pw[k] after k iterations
            out[sp + "w_end"] = st["w_end"]; out[sp + "p_end"] = st["p_end"]; out[sp + m + "G_end"] = st["G_end"]
            out[sp + "tr_end"] = st["tr_end"]; out[sp + "grad_end"] = st["grad_end"]
    return out


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote %-40s %7.1f KB" % (os.path.basename(path), os.path.getsize(path) / 1024))


def loader_fixture():
    """Pin load_csv_dataset (riemannhamiltonianmontecarlo_amd/data.py) to the reference driver's own preprocessing: the block
    main.py:19-41 (dataset_name .. XX = np.hstack) is EXECUTED from the reference file, unmodified apart from the dataset name, with
    cwd = code/ as the script expects, for the two data sets it can select ('australian', 'heart': main.py:22,28).  Stored: the raw
    CSV values and the (XX, t) the reference's lines produced."""
    src = open(os.path.join(REF, "main.py")).read().splitlines()
    assert src[17].strip() == "if __name__ == '__main__':" and src[18].strip().startswith("#%% Load and preprocess data"), src[17:19]
    assert src[40].strip() == "XX = np.hstack((XX, X))", src[40]
    block = "\n".join(line[4:] for line in src[19:41])          # lines 20-41, one indent level removed
    out = {}
    cwd = os.getcwd()
    os.chdir(REF)
    try:
        for ds in ("australian", "heart"):
            code = block.replace("dataset_name = 'australian'", "dataset_name = %r" % ds)
            assert code.count("dataset_name = %r" % ds) == 1
            env = {"np": np}
            exec(compile(code, "main.py[20:41]", "exec"), env)
            out[ds + "_raw"] = np.loadtxt(os.path.join(REF, "data", ds + ".csv"), delimiter=",")
            out[ds + "_XX"] = np.asarray(env["XX"], dtype=np.float64)
            out[ds + "_t"] = np.asarray(env["t"], dtype=np.float64)
    finally:
        os.chdir(cwd)
    save("loader_main_py", **out)


def ripley_tape():
    """RMHMC transitions of the reference on Ripley's data with the authors' cubic basis [1, X, X^2, X^3] (D = 7,
    authors_code/Bayes_Log_Reg/MCMC/BLR_RMHMC.m:151-173): the Python reference is fed that XX (its main.py cannot select ripley)."""
    XX, t = load_csv_dataset(os.path.join(REF, "data", "ripley.csv"), polynomial_order=3)
    save("tape_ripley", **capture(XX, t, 12, 40))


def ess_4096_fixture():
    """tools.CalculateESS / tools.ac at S = 4096, a power of two: nFFT = nextpow2(S)+1 = 4097 (tools.py:23), so the circular
    autocorrelation wraps from lag 1 on - the case where the reference's Python FFT length and the MATLAB one (ac.m:78) differ most."""
    rs = np.random.RandomState(321)
    S, P = 4096, 2
    x = np.zeros((S, P)); rho = np.array([0.5, 0.97]); e = rs.randn(S, P)
    for i in range(1, S):
        x[i] = rho * x[i - 1] + e[i]
    ess = ref_tools.CalculateESS(x, S - 1)
    acf = np.stack([ref_tools.ac(x[:, j], 64) for j in range(P)], axis=1)
    save("ess_s4096", samples=x.astype(np.float64), ess=ess.ravel(), acf64=acf, maxlag=np.int64(S - 1))


def large_d_tapes():
    """64 < D <= 256: the blocked (tiled-Cholesky) path of the build, pinned to rmhmc.py:96-175 itself.  D = 130 with the
    matrices of the first transition in full (three column blocks, ragged M); BASELINE config 5's own shape (M = 50000,
    D = 256: the reference forms its 134 MB InvGdG tensor twice, about a minute here) for exactly one leapfrog step with
    vectors, scalars and diagonals only."""
    for name, M, D, dseed, seed, n_iter, L, compact in (
            ("syn_m3001_d130", 3001, 130, 4, 10, 3, 2, False),
            ("syn_m50000_d256_L1", 50000, 256, 0, 11, 1, 1, True)):
        XX, t = synthetic_logreg(M, D, dseed)
        g = capture(XX, t, seed, n_iter, L=L, detail_iters=1, compact=compact)
        if not compact:      # keep G0 and the per-step G_end of the detailed transition, drop the three other D x D copies
            for k in ("it0_InvG0", "it0_cholG0", "it0_G_prop"):
                g.pop(k)
        g.update(M=np.int64(M), D=np.int64(D), data_seed=np.int64(dseed))
        save("tape_" + name, **g)


def main():
    _check_lines()
    groups = set(sys.argv[1:]) or {"base", "ripley", "loader", "ess4096", "larged"}
    if "larged" in groups:
        large_d_tapes()
    if "ripley" in groups:
        ripley_tape()
    if "loader" in groups:
        loader_fixture()
    if "ess4096" in groups:
        ess_4096_fixture()
    if "base" not in groups:
        return
    # --- bundled datasets (main.py:20-41 preprocessing), stored as data fixtures ------------------
    for ds in ("pima", "australian", "german", "heart"):
        XX, t = load_csv_dataset(os.path.join(REF, "data", ds + ".csv"))
        save("data_" + ds, XX=XX, t=t)
    XX, t = load_csv_dataset(os.path.join(REF, "data", "ripley.csv"), polynomial_order=3)  # cubic basis, D = 7
    save("data_ripley", XX=XX, t=t)
    # --- transition tapes on the bundled data ------------------------------------------------------
    for ds, seed, n_iter in (("pima", 1, 40), ("australian", 2, 30), ("german", 3, 10), ("heart", 4, 24)):
        XX, t = load_csv_dataset(os.path.join(REF, "data", ds + ".csv"))
        save("tape_" + ds, **capture(XX, t, seed, n_iter))
    # --- synthetic recipes (X, t regenerated from the seed, not stored) -----------------------------
    for name, M, D, dseed, seed, n_iter, L in (
            ("syn_m1000_d8", 1000, 8, 0, 5, 30, 6),       # BASELINE config 2 shape
            ("syn_m50_d5_L1", 50, 5, 1, 6, 30, 1),        # exactly one leapfrog step per transition
            ("syn_m300_d20", 300, 20, 2, 7, 12, 6),
            ("syn_m203_d33", 203, 33, 3, 8, 6, 3),        # ragged: M not a multiple of 4, D not of 16
            ("syn_m10000_d64_L1", 10000, 64, 0, 9, 2, 1)):  # BASELINE config 3 shape; momentum guard fires
        XX, t = synthetic_logreg(M, D, dseed)
        g = capture(XX, t, seed, n_iter, L=L, detail_iters=1 if D >= 64 else 2)
        g.update(M=np.int64(M), D=np.int64(D), data_seed=np.int64(dseed))
        save("tape_" + name, **g)
    # --- a case where the position guard (rmhmc.py:125-130) fires and the state stays finite -------
    XX, t = synthetic_logreg(40, 3, 11)
    for seed in range(200):
        g = capture(XX * 6.0, t, seed, 6, L=6, eps=0.9, detail_iters=0)
        if g["guard_w_fired"] > 0 and np.all(np.isfinite(g["w_prop"])) and np.all(np.isfinite(g["H_prop"])):
            g.update(M=np.int64(40), D=np.int64(3), data_seed=np.int64(11), x_scale=np.float64(6.0))
            save("tape_guard_w", **g)
            print("  position guard fired %d times with seed %d" % (g["guard_w_fired"], seed))
            break
    else:
        print("  WARNING: no finite position-guard case found")
    # --- ESS (tools.py:21-74) ------------------------------------------------------------------------
    rs = np.random.RandomState(123)
    S, P = 600, 3
    x = np.zeros((S, P))
    rho = np.array([0.0, 0.6, 0.95])
    e = rs.randn(S, P)
    for i in range(1, S):
        x[i] = rho * x[i - 1] + e[i]
    ess = ref_tools.CalculateESS(x, S - 1)
    acf = np.stack([ref_tools.ac(x[:, j], S - 1) for j in range(P)], axis=1)
    save("ess_ar1", samples=x, ess=ess.ravel(), acf=acf, maxlag=np.int64(S - 1))
    XX, t = load_csv_dataset(os.path.join(REF, "data", "pima.csv"))
    np.random.seed(21)
    with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
        smp, _ = ref_rmhmc.RMHMC(XX, t, NumOfIterations=400, BurnIn=100)
    smp = smp[1:]  # row 0 is never written by the reference (np.empty)
    ess = ref_tools.CalculateESS(smp, smp.shape[0] - 1)
    save("ess_pima_chain", samples=smp, ess=ess.ravel(), maxlag=np.int64(smp.shape[0] - 1))


if __name__ == "__main__":
    main()
