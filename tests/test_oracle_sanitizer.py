"""AddressSanitizer + UBSan run of the CPU oracle (GPU ASan is not available on the pool, so the sanitizer
coverage of the native code is the CPU restatement)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def test_oracle_under_asan_ubsan():
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "asan_oracle_run.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan run ok" in r.stdout, r.stderr[-2000:]
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr
