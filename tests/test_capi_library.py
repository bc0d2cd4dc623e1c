"""CPU-side checks of the product library: it loads, exports every symbol include/rmhmc.h declares,
and fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from riemannhamiltonianmontecarlo_amd import _capi, RMHMC


def _declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "rmhmc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rmhmc_[a-z0-9_]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    syms = _declared_symbols()
    assert len(syms) >= 15
    assert set(syms) == set(_capi.SIGNATURES), set(syms) ^ set(_capi.SIGNATURES)


def test_hip_library_exports_every_declared_symbol(hip):
    import ctypes
    lib = ctypes.CDLL(hip.path)
    for s in _declared_symbols():
        assert hasattr(lib, s), s
    assert "gfx950" in hip.version()


def test_oracle_exports_the_same_abi(oracle):
    import ctypes
    lib = ctypes.CDLL(oracle.path)
    for s in _declared_symbols():
        assert hasattr(lib, s), s


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(hip):
    with pytest.raises(_capi.RmhmcError) as e:
        hip.context(10, 3, 1)
    assert e.value.code == -2 and "no HIP device" in str(e.value)
    X = np.random.RandomState(0).randn(10, 3); t = (X[:, 0] > 0).astype(float)
    with pytest.raises(_capi.RmhmcError):
        RMHMC(X, t, NumOfIterations=4, BurnIn=1, verbose=False)


def test_shape_limits_rejected_before_any_device_call(hip):
    """rmhmc_create's shape checks need no GPU: D > 256, and a D <= 64 data matrix of 4 GB or more (32-bit offsets of the row passes)."""
    for shape in ((100, 257, 1), ((1 << 23) + 1, 64, 1), (1 << 25, 16, 1)):
        with pytest.raises(_capi.RmhmcError) as e:
            hip.context(*shape)
        assert e.value.code == -4, (shape, str(e.value))  # RMHMC_ERR_UNSUPPORTED
    assert "4 GB" in str(e.value)


def test_shim_argument_checks():
    X = np.zeros((5, 2)); t = np.zeros(5)
    with pytest.raises(ValueError):
        RMHMC(X, t, NumOfIterations=10, BurnIn=10)
    with pytest.raises(ValueError):
        RMHMC(X, np.zeros(4))


def test_empty_inputs_rejected(oracle):
    for bad in ((0, 3, 1), (10, 0, 1), (10, 3, 0)):
        with pytest.raises(_capi.RmhmcError) as e:
            oracle.context(*bad)
        assert e.value.code == -1
    with oracle.context(5, 2, 1) as ctx:
        with pytest.raises(ValueError):
            ctx.set_data(np.zeros((4, 2)), np.zeros(4))
        with pytest.raises(ValueError):
            ctx.sample(5, 5)


def test_int8_metric_flag_helpers():
    """host logic of the int8 matrix-core option (include/rmhmc.h: RMHMC_FLAG_INT8_METRIC, slices in bits 12..14)"""
    from riemannhamiltonianmontecarlo_amd import _capi
    assert _capi.int8_metric_flags(6) == (1 << 5) | (6 << 12)
    assert _capi.int8_metric_flags(4) & _capi.FLAG_INT8_METRIC
    for bad in (3, 8, 0):
        with pytest.raises(ValueError):
            _capi.int8_metric_flags(bad)
    # auto rule: 6 slices where the path applies and the batch fills its 128-chain tiles, else the fp64 matrix cores
    # (chosen by the shim => RMHMC_FLAG_INT8_CERTIFY: rmhmc_set_data may send data it cannot certify to 1e-9 to the fp64 kernels)
    auto6 = _capi.int8_metric_flags(6) | _capi.FLAG_INT8_CERTIFY
    assert _capi.FLAG_INT8_CERTIFY == 1 << 7
    assert _capi.auto_metric_flags(64, 8192) == auto6
    assert _capi.auto_metric_flags(256, 4096) == auto6
    assert _capi.auto_metric_flags(64, 100) == 0 and _capi.auto_metric_flags(8, 8192) == 0 and _capi.auto_metric_flags(300, 8192) == 0
    assert _capi.auto_metric_flags(64, 100, 5) == _capi.int8_metric_flags(5) and _capi.auto_metric_flags(64, 8192, 0) == 0
    # with the data size known the rule is the amount of work, chains * M * D^2 >= 1e9
    assert _capi.auto_metric_flags(64, 128, M=10000) == auto6
    assert _capi.auto_metric_flags(15, 8192, M=690) == auto6
    assert _capi.auto_metric_flags(25, 600, M=1000) == 0 and _capi.auto_metric_flags(15, 1, M=690) == 0
    # the flag values of the header and of the binding agree
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "rmhmc.h")).read()
    assert re.search(r"#define RMHMC_FLAG_INT8_METRIC \(1u << 5\)", hdr)
    assert re.search(r"#define RMHMC_FLAG_INT8_SLICES\(S\) \(\(\(uint32_t\)\(S\) & 7u\) << 12\)", hdr)
    # every single-bit flag of the header has the same value in the binding
    for name, bit in re.findall(r"#define RMHMC_(FLAG_[A-Z0-9_]+) \(1u << (\d+)\)", hdr):
        assert getattr(_capi, name) == 1 << int(bit), name


def test_build_tracks_every_included_header():
    """build() rebuilds librmhmc_hip.so when ANY header the translation unit pulls in changes (two hot-path headers
    were missing from a hand-written list).  Every #include "..." reachable from rmhmc_hip.hip must be in hip_sources()."""
    import __graft_entry__ as ge
    srcs = {os.path.realpath(s) for s in ge.hip_sources()}
    todo, seen = [os.path.join(ge.CSRC, "rmhmc_hip.hip")], set()
    while todo:
        f = todo.pop()
        if f in seen:
            continue
        seen.add(f)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(f).read(), flags=re.M):
            path = os.path.realpath(os.path.join(os.path.dirname(f), inc))
            assert path in srcs, "%s (included by %s) is not tracked by build()" % (inc, os.path.basename(f))
            todo.append(path)
    assert len(seen) >= 6
