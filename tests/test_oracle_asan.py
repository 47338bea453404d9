"""CPU: the oracle's unit tests once more against an AddressSanitizer + UBSan build of the oracle
(oracle/Makefile `asan`; SURVEY.md section 5: the reference has live UB that the oracle must not inherit).
GPU sanitizers are not available on the pool, so this is the sanitizer coverage of the build."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _runtime(name):
    out = subprocess.run(["gcc", "-print-file-name=" + name], capture_output=True, text=True)
    path = out.stdout.strip()
    return path if os.path.isabs(path) and os.path.exists(path) else None


def test_oracle_units_under_asan_ubsan():
    asan = _runtime("libasan.so")
    if asan is None:
        pytest.skip("libasan.so not installed with this gcc")
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, capture_output=True)
    lib = os.path.join(ROOT, "oracle", "_build", "libvis_oracle_asan.so")
    assert os.path.exists(lib)
    env = dict(os.environ)
    ubsan = _runtime("libubsan.so")
    env["LD_PRELOAD"] = asan + ((":" + ubsan) if ubsan else "")
    # python itself "leaks" at exit; any other finding aborts the child with a non-zero exit code
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=1:halt_on_error=1"
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    env["VIS_ORACLE_LIB"] = lib
    units = [os.path.join(ROOT, "tests", f) for f in ("test_oracle_units.py", "test_align_oracle.py") if os.path.exists(os.path.join(ROOT, "tests", f))]
    # -s: a sanitizer report must reach our pipe before the abort
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-p", "no:cacheprovider"] + units,
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "runtime error" not in out.stdout + out.stderr, tail          # UBSan reports
    assert "passed" in out.stdout, tail
