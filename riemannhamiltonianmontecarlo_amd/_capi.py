"""ctypes binding of the C-ABI declared in include/rmhmc.h.

``RmhmcLib(path)`` binds any shared library exporting that ABI.  The product
uses it with the HIP library only (``load_hip_library``), which raises if the
extension is missing or cannot run — there is no CPU fallback.  The test-suite
binds the CPU oracle with the same class to diff results.
"""
import ctypes as C
import os

import numpy as np

F64 = 0
FLAG_MOMENTUM_LT = 1 << 0
FLAG_GUARDS = 1 << 1
COMPAT = FLAG_MOMENTUM_LT | FLAG_GUARDS
FLAG_FP32_METRIC = 1 << 4
FLAG_INT8_METRIC = 1 << 5
FLAG_MMALA_FULL = 1 << 6
FLAG_INT8_CERTIFY = 1 << 7
FLAG_INT8_INNER_FULL = 1 << 10   # int8 x 6 slices: position iterates before the last also from all 6 slices (default: 5)
FLAG_ESS_WRAP = 1 << 9      # ESS with the reference Python's wrapped FFT length (tools.py:23); default = MATLAB (no wrap)
INT8_CERTIFY_TOL = 1e-9


def int8_metric_flags(slices=6):
    """flags for the int8 matrix-core metric assembly with `slices` byte slices per operand (4..7), include/rmhmc.h"""
    if not 4 <= int(slices) <= 7:
        raise ValueError("slices must be 4..7")
    return FLAG_INT8_METRIC | (int(slices) << 12)


def auto_metric_flags(D, n_chains, slices=None, M=None):
    """None: 6 slices where the int8 path pays: 8 < D <= 256 and enough work to fill its 128 x 128 tiles, n_chains * M * D^2 >= 1e9
    (measured, tools/i8_threshold.py: 1.3-1.7x over the fp64 matrix cores from 128 chains x 10000 rows x D 64 and from 8192 chains of
    the 690 x 15 australian data upwards; 0.76x at 600 chains x 1000 x 25), or n_chains >= 1024 when M is not given; 0: fp64 cores"""
    auto = slices is None
    if slices is None:
        big = (n_chains * float(M) * D * D >= 1e9) if M is not None else (n_chains >= 1024)
        slices = 6 if (8 < D <= 256 and big) else 0
    if not slices:
        return 0
    # chosen by the shim, not by the caller: let rmhmc_set_data check the error bound for the actual data and fall back to fp64
    return int8_metric_flags(slices) | (FLAG_INT8_CERTIFY if auto else 0)


FLAG_ORACLE_LITERAL = 1 << 8

ST_NOT_PD, ST_NONFINITE, ST_GUARD_P, ST_GUARD_W = 1, 2, 4, 8

_PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.path.join(_PKG, "csrc", "librmhmc_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)

PROGRESS_FN = C.CFUNCTYPE(None, C.c_int32, C.c_int64, C.c_int64, C.c_int64, C.c_void_p)
EV_PROGRESS, EV_BURNIN_DONE = 0, 1



class Option(C.Structure):
    """rmhmc_option of include/rmhmc.h: one (key, value) tuning option"""
    _fields_ = [("key", C.c_char_p), ("value", C.c_int64)]


# every symbol include/rmhmc.h declares, with its signature
SIGNATURES = {
    "rmhmc_create_opts": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_uint32,
                                    C.POINTER(Option), C.c_int32]),
    "rmhmc_set_option": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "rmhmc_get_option": (C.c_int, [C.c_void_p, C.c_char_p, _lp]),
    "rmhmc_options": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rmhmc_version": (C.c_char_p, []),
    "rmhmc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int64, C.c_int32, C.c_int64, C.c_int32, C.c_uint32]),
    "rmhmc_destroy": (None, [C.c_void_p]),
    "rmhmc_last_error": (C.c_char_p, [C.c_void_p]),
    "rmhmc_device_info": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "rmhmc_set_data": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double]),
    "rmhmc_log_posterior": (C.c_int, [C.c_void_p, _dp, _dp]),
    "rmhmc_metric": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "rmhmc_metric_terms": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp]),
    "rmhmc_leapfrog": (C.c_int, [C.c_void_p, _dp, _dp, C.c_double, _ip, _ip, C.c_int32, _dp, _ip]),
    "rmhmc_transition": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, C.c_int32,
                                   _ip, _ip, _dp, _dp, _dp, _dp, _dp, _ip]),
    "rmhmc_sample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_uint64,
                               C.c_int64, _dp, _dp, _lp, _lp, _dp]),
    "rmhmc_chains_init": (C.c_int, [C.c_void_p, _dp, C.c_uint64, C.c_int64, C.c_int32, C.c_double, C.c_int32]),
    "rmhmc_chains_run": (C.c_int, [C.c_void_p, C.c_int64]),
    "rmhmc_chains_state": (C.c_int, [C.c_void_p, _dp, _lp, _lp]),
    "rmhmc_chains_restore": (C.c_int, [C.c_void_p, _lp, _lp]),
    "rmhmc_kernel_time": (C.c_int, [C.c_void_p, C.c_char_p, _dp, _lp]),
    "rmhmc_int8_certificate": (C.c_int, [C.c_void_p, _dp, _ip]),
    "rmhmc_set_progress": (C.c_int, [C.c_void_p, PROGRESS_FN, C.c_int64, C.c_int64, C.c_void_p]),
    "rmhmc_ess": (C.c_int, [C.c_void_p, _dp, C.c_int64, C.c_int64, C.c_int32, _dp]),
    "rmhmc_sample_stats": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_uint64, C.c_int64,
                                     _dp, _dp, _dp, _dp, _lp, _lp, _dp]),
    "rmhmc_chains_state_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rmhmc_sample_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_uint64,
                                   C.c_int64, _dp, C.c_void_p, C.c_void_p, C.c_void_p, _dp]),
    "rmhmc_sample_stats_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_int32, C.c_uint64, C.c_int64,
                                         _dp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, _dp]),
    "rmhmc_mmala_transition": (C.c_int, [C.c_void_p, _dp, _dp, _dp, C.c_double, _ip, _dp, _dp]),
    "rmhmc_mmala_sample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_double, C.c_uint64, C.c_int64, _dp, _dp, _lp, _dp]),
    "rmhmc_hmc_transition": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, C.c_int32, C.c_double, _ip, _ip, _dp, _dp, _dp, _dp]),
    "rmhmc_hmc_sample": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_double, C.c_uint64, C.c_int64, _dp, _dp,
                                   _lp, _lp, _dp]),
}


class RmhmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("rmhmc error %d: %s" % (code, msg))
        self.code = code


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _ptr(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


class RmhmcLib:
    """A loaded library exporting the rmhmc C-ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise FileNotFoundError("rmhmc library not built: %s (run `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
        self.path = path
        self.lib = C.CDLL(path, mode=C.RTLD_LOCAL)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(self.lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        # the HIP library writes its _dev outputs through device pointers of its own GPU; the CPU oracle's "device" is the host
        self.on_gpu = "gfx950" in self.version()

    def version(self):
        return self.lib.rmhmc_version().decode()

    def context(self, M, D, n_chains=1, flags=COMPAT, device=0, options=None):
        """options: dict of the tuning options of include/rmhmc.h (rmhmc_create_opts), e.g. {"graph": 0, "medium": 0}"""
        return Context(self, M, D, n_chains, flags, device, options)


class Context:
    """One opaque rmhmc_ctx: M data rows, D dims, n_chains chains on one device."""

    def __init__(self, rl, M, D, n_chains, flags, device, options=None):
        self.rl, self.lib = rl, rl.lib
        self.M, self.D, self.n, self.flags = int(M), int(D), int(n_chains), int(flags)
        self._h = C.c_void_p()
        if options:
            arr = (Option * len(options))(*[Option(str(k).encode(), int(v)) for k, v in options.items()])
            rc = self.lib.rmhmc_create_opts(C.byref(self._h), device, self.M, self.D, self.n, F64, self.flags, arr, len(options))
        else:
            rc = self.lib.rmhmc_create(C.byref(self._h), device, self.M, self.D, self.n, F64, self.flags)
        if rc != 0:
            raise RmhmcError(rc, self.lib.rmhmc_last_error(None).decode())

    def close(self):
        if self._h:
            self.lib.rmhmc_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise RmhmcError(rc, self.lib.rmhmc_last_error(self._h).decode())

    def device_info(self):
        buf = C.create_string_buffer(1024)
        self._ck(self.lib.rmhmc_device_info(self._h, buf, 1024))
        return buf.value.decode()

    def set_option(self, key, value):
        """a run-time tuning option (include/rmhmc.h); create-time ones go to ``context(..., options={...})``"""
        self._ck(self.lib.rmhmc_set_option(self._h, str(key).encode(), int(value)))

    def get_option(self, key):
        v = C.c_int64(0)
        self._ck(self.lib.rmhmc_get_option(self._h, str(key).encode(), C.cast(C.byref(v), _lp)))
        return v.value

    def options(self):
        """the active option set as a dict (empty for the CPU oracle)"""
        buf = C.create_string_buffer(1024)
        self._ck(self.lib.rmhmc_options(self._h, buf, 1024))
        return {k: int(v) for k, v in (kv.split("=") for kv in buf.value.decode().split())}

    def set_data(self, XX, t, alpha=100.0):
        XX = _f64(XX)
        t = _f64(t).ravel()
        if XX.shape != (self.M, self.D) or t.shape != (self.M,):
            raise ValueError("XX must be (%d,%d) and t (%d,)" % (self.M, self.D, self.M))
        self._ck(self.lib.rmhmc_set_data(self._h, _ptr(XX), _ptr(t), float(alpha)))

    # ---- unit entry points -------------------------------------------------
    def log_posterior(self, w):
        w = _f64(w, (self.n, self.D))
        out = np.empty(self.n)
        self._ck(self.lib.rmhmc_log_posterior(self._h, _ptr(w), _ptr(out)))
        return out

    def metric(self, w):
        w = _f64(w, (self.n, self.D))
        G = np.empty((self.n, self.D, self.D)); hld = np.empty(self.n); g = np.empty((self.n, self.D))
        self._ck(self.lib.rmhmc_metric(self._h, _ptr(w), _ptr(G), _ptr(hld), _ptr(g)))
        return G, hld, g

    def metric_terms(self, w, p=None):
        w = _f64(w, (self.n, self.D))
        tr = np.empty((self.n, self.D))
        q = None
        if p is not None:
            p = _f64(p, (self.n, self.D))
            q = np.empty((self.n, self.D))
        self._ck(self.lib.rmhmc_metric_terms(self._h, _ptr(w), _ptr(p), _ptr(tr), _ptr(q)))
        return tr, q

    def leapfrog(self, w, p, eps, direction, nsteps, K=4):
        w = _f64(w, (self.n, self.D)).copy(); p = _f64(p, (self.n, self.D)).copy()
        d = np.ascontiguousarray(np.broadcast_to(direction, (self.n,)), dtype=np.int32)
        ns = np.ascontiguousarray(np.broadcast_to(nsteps, (self.n,)), dtype=np.int32)
        hld = np.empty(self.n); st = np.zeros(self.n, dtype=np.int32)
        self._ck(self.lib.rmhmc_leapfrog(self._h, _ptr(w), _ptr(p), float(eps), _ptr(d, _ip), _ptr(ns, _ip), int(K),
                                         _ptr(hld), _ptr(st, _ip)))
        return w, p, hld, st

    def transition(self, w, z, u_len, g_dir, u_acc, L=6, eps=0.5, K=4):
        n, D = self.n, self.D
        w = _f64(w, (n, D)).copy()
        z = _f64(z, (n, D)); u_len = _f64(u_len, (n,)); g_dir = _f64(g_dir, (n,)); u_acc = _f64(u_acc, (n,))
        acc = np.zeros(n, dtype=np.int32); ns = np.zeros(n, dtype=np.int32); st = np.zeros(n, dtype=np.int32)
        Hc = np.empty(n); Hp = np.empty(n); wp = np.empty((n, D)); pp = np.empty((n, D)); hp = np.empty(n)
        self._ck(self.lib.rmhmc_transition(self._h, _ptr(w), _ptr(z), _ptr(u_len), _ptr(g_dir), _ptr(u_acc), int(L),
                                           float(eps), int(K), _ptr(acc, _ip), _ptr(ns, _ip), _ptr(Hc), _ptr(Hp),
                                           _ptr(wp), _ptr(pp), _ptr(hp), _ptr(st, _ip)))
        return dict(w=w, accepted=acc, nsteps=ns, H_cur=Hc, H_prop=Hp, w_prop=wp, p_prop=pp, hld_prop=hp, status=st)

    # ---- bulk entry points -------------------------------------------------
    def sample(self, n_iter, burn_in, L=6, eps=0.5, K=4, seed=0, chain_offset=0, theta0=None):
        n, D = self.n, self.D
        S = int(n_iter) - int(burn_in)
        if S <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        samples = np.empty((n, S, D)); acc = np.zeros(n, dtype=np.int64); steps = np.zeros(n, dtype=np.int64)
        secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_sample(self._h, int(n_iter), int(burn_in), int(L), float(eps), int(K), int(seed),
                                       int(chain_offset), _ptr(th), _ptr(samples), _ptr(acc, _lp), _ptr(steps, _lp),
                                       C.cast(C.byref(secs), _dp)))
        return samples, acc, steps, secs.value

    # ---- ESS on the device (tools.py:32-74) -----------------------------------
    def ess(self, samples):
        samples = _f64(samples)
        if samples.ndim == 2:
            samples = samples[None]
        n, S, P = samples.shape
        out = np.empty((n, P))
        self._ck(self.lib.rmhmc_ess(self._h, _ptr(samples), n, S, P, _ptr(out)))
        return out

    def sample_stats(self, n_iter, burn_in, L=6, eps=0.5, K=4, seed=0, chain_offset=0, theta0=None):
        n, D = self.n, self.D
        if int(n_iter) - int(burn_in) <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        mean = np.empty((n, D)); var = np.empty((n, D)); ess = np.empty((n, D))
        acc = np.zeros(n, dtype=np.int64); steps = np.zeros(n, dtype=np.int64)
        secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_sample_stats(self._h, int(n_iter), int(burn_in), int(L), float(eps), int(K), int(seed),
                                             int(chain_offset), _ptr(th), _ptr(mean), _ptr(var), _ptr(ess), _ptr(acc, _lp),
                                             _ptr(steps, _lp), C.cast(C.byref(secs), _dp)))
        return dict(mean=mean, var=var, ess=ess, accepted=acc, leapfrog_steps=steps, seconds=secs.value)

    # ---- device-resident write-out: outputs are torch tensors on the context's GPU (host tensors for the CPU oracle) ----------
    def _out_tensors(self, shapes, device):
        """device: a torch.device.  The HIP library writes through device pointers on its own GPU; the oracle's "device" is the host."""
        import torch
        return [torch.empty(shape, dtype=dt, device=device) for shape, dt in shapes]

    def chains_state_dev(self, device):
        import torch
        w, it, acc = self._out_tensors([((self.n, self.D), torch.float64), ((self.n,), torch.int64), ((self.n,), torch.int64)], device)
        self._ck(self.lib.rmhmc_chains_state_dev(self._h, w.data_ptr(), it.data_ptr(), acc.data_ptr()))
        return w, it, acc

    def sample_dev(self, device, n_iter, burn_in, L=6, eps=0.5, K=4, seed=0, chain_offset=0, theta0=None):
        import torch
        n, D = self.n, self.D
        S = int(n_iter) - int(burn_in)
        if S <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        smp, acc, steps = self._out_tensors([((n, S, D), torch.float64), ((n,), torch.int64), ((n,), torch.int64)], device)
        secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_sample_dev(self._h, int(n_iter), int(burn_in), int(L), float(eps), int(K), int(seed), int(chain_offset),
                                           _ptr(th), smp.data_ptr(), acc.data_ptr(), steps.data_ptr(), C.cast(C.byref(secs), _dp)))
        return smp, acc, steps, secs.value

    def sample_stats_dev(self, device, n_iter, burn_in, L=6, eps=0.5, K=4, seed=0, chain_offset=0, theta0=None):
        import torch
        n, D = self.n, self.D
        if int(n_iter) - int(burn_in) <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        mean, var, ess, acc, steps = self._out_tensors([((n, D), torch.float64)] * 3 + [((n,), torch.int64)] * 2, device)
        secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_sample_stats_dev(self._h, int(n_iter), int(burn_in), int(L), float(eps), int(K), int(seed), int(chain_offset),
                                                 _ptr(th), mean.data_ptr(), var.data_ptr(), ess.data_ptr(), acc.data_ptr(), steps.data_ptr(),
                                                 C.cast(C.byref(secs), _dp)))
        return dict(mean=mean, var=var, ess=ess, accepted=acc, leapfrog_steps=steps, seconds=secs.value)

    # ---- simplified mMALA (authors_code/.../BLR_mMALA_Simp.m) -------------------
    def mmala_transition(self, w, z, u_acc, eps=1.0):
        n, D = self.n, self.D
        w = _f64(w, (n, D)).copy(); z = _f64(z, (n, D)); u_acc = _f64(u_acc, (n,))
        acc = np.zeros(n, dtype=np.int32); ratio = np.empty(n); wp = np.empty((n, D))
        self._ck(self.lib.rmhmc_mmala_transition(self._h, _ptr(w), _ptr(z), _ptr(u_acc), float(eps), _ptr(acc, _ip), _ptr(ratio), _ptr(wp)))
        return dict(w=w, accepted=acc, ratio=ratio, w_prop=wp)

    def mmala_sample(self, n_iter, burn_in, eps=1.0, seed=0, chain_offset=0, theta0=None):
        n, D = self.n, self.D
        S = int(n_iter) - int(burn_in)
        if S <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        samples = np.empty((n, S, D)); acc = np.zeros(n, dtype=np.int64); secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_mmala_sample(self._h, int(n_iter), int(burn_in), float(eps), int(seed), int(chain_offset), _ptr(th),
                                             _ptr(samples), _ptr(acc, _lp), C.cast(C.byref(secs), _dp)))
        return samples, acc, secs.value

    # ---- plain HMC (code/hmc.py) ---------------------------------------------
    def hmc_transition(self, w, z, u_len, u_acc, L=100, eps=0.14):
        n, D = self.n, self.D
        w = _f64(w, (n, D)).copy()
        z = _f64(z, (n, D)); u_len = _f64(u_len, (n,)); u_acc = _f64(u_acc, (n,))
        acc = np.zeros(n, dtype=np.int32); ns = np.zeros(n, dtype=np.int32)
        Hc = np.empty(n); Hp = np.empty(n); wp = np.empty((n, D)); pp = np.empty((n, D))
        self._ck(self.lib.rmhmc_hmc_transition(self._h, _ptr(w), _ptr(z), _ptr(u_len), _ptr(u_acc), int(L), float(eps),
                                               _ptr(acc, _ip), _ptr(ns, _ip), _ptr(Hc), _ptr(Hp), _ptr(wp), _ptr(pp)))
        return dict(w=w, accepted=acc, nsteps=ns, H_cur=Hc, H_prop=Hp, w_prop=wp, p_prop=pp)

    def hmc_sample(self, n_iter, burn_in, L=100, eps=0.14, seed=0, chain_offset=0, theta0=None):
        n, D = self.n, self.D
        S = int(n_iter) - int(burn_in)
        if S <= 0:
            raise ValueError("BurnIn must be < NumOfIterations")
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (n, D)))
        samples = np.empty((n, S, D)); acc = np.zeros(n, dtype=np.int64); steps = np.zeros(n, dtype=np.int64)
        secs = C.c_double(0.0)
        self._ck(self.lib.rmhmc_hmc_sample(self._h, int(n_iter), int(burn_in), int(L), float(eps), int(seed),
                                           int(chain_offset), _ptr(th), _ptr(samples), _ptr(acc, _lp), _ptr(steps, _lp),
                                           C.cast(C.byref(secs), _dp)))
        return samples, acc, steps, secs.value

    def chains_init(self, theta0=None, seed=0, chain_offset=0, L=6, eps=0.5, K=4):
        th = None if theta0 is None else _f64(np.broadcast_to(theta0, (self.n, self.D)))
        self._ck(self.lib.rmhmc_chains_init(self._h, _ptr(th), int(seed), int(chain_offset), int(L), float(eps), int(K)))

    def chains_run(self, n_steps):
        self._ck(self.lib.rmhmc_chains_run(self._h, int(n_steps)))

    def chains_state(self):
        w = np.empty((self.n, self.D)); it = np.zeros(self.n, dtype=np.int64); acc = np.zeros(self.n, dtype=np.int64)
        self._ck(self.lib.rmhmc_chains_state(self._h, _ptr(w), _ptr(it, _lp), _ptr(acc, _lp)))
        return w, it, acc

    def chains_restore(self, iters, accepted):
        it = np.ascontiguousarray(iters, dtype=np.int64).reshape(self.n)
        acc = np.ascontiguousarray(accepted, dtype=np.int64).reshape(self.n)
        self._ck(self.lib.rmhmc_chains_restore(self._h, _ptr(it, _lp), _ptr(acc, _lp)))

    def set_progress(self, fn, first=49, every=50):
        """fn(event, iterations_done, accepted_total, iterations_total) or None.  The defaults are the reference's schedule: a report
        whenever IterationNum+1 is a multiple of 50, i.e. after 49, 99, ... completed transitions (rmhmc.py:38).  One chain: exactly
        then; several chains: when the slowest chain has got there (include/rmhmc.h)."""
        if fn is None:
            self._progress_cb = PROGRESS_FN(0)
            self._ck(self.lib.rmhmc_set_progress(self._h, self._progress_cb, 1, 1, None))
            return
        self._progress_cb = PROGRESS_FN(lambda ev, it, acc, itot, user: fn(int(ev), int(it), int(acc), int(itot)))   # kept alive with the context
        self._ck(self.lib.rmhmc_set_progress(self._h, self._progress_cb, int(first), int(every), None))

    def int8_certificate(self):
        """(bound, active): worst-case error bound of the int8 metric path for the current data, and whether the int8 kernels are in use"""
        b = C.c_double(0.0); a = C.c_int32(0)
        self._ck(self.lib.rmhmc_int8_certificate(self._h, C.cast(C.byref(b), _dp), C.cast(C.byref(a), _ip)))
        return b.value, bool(a.value)

    def kernel_time(self, which):
        s = C.c_double(0.0); k = C.c_int64(0)
        self._ck(self.lib.rmhmc_kernel_time(self._h, which.encode(), C.cast(C.byref(s), _dp), C.cast(C.byref(k), _lp)))
        return s.value, k.value


_hip = None


def load_hip_library():
    """The product library (HIP kernels for gfx950).  Raises when it is not built.

    PyTorch (used for device plumbing and torch.distributed) bundles its own HIP runtime with the
    same SONAME as the system one.  Importing torch BEFORE dlopen-ing the library makes both share
    that single runtime; the other order would put two HIP/HSA stacks in one process."""
    global _hip
    if _hip is None:
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        _hip = RmhmcLib(os.environ.get("RMHMC_HIP_LIB", HIP_LIB_PATH))   # (override: experiment builds of the same library)
    return _hip
