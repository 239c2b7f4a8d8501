"""The oracle under AddressSanitizer + UBSan (CPU only; GPU sanitizers are not available on the pool).
The reference has real memory bugs on this path (1-element F with grad-diff, bak buffers
read before they are written); the restatement must not -- its traces are what the HIP path is held to."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    libasan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "oracle", "liboracle_asan.so")
    env = dict(os.environ, ORACLE_SO=so, LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                          os.path.join(ROOT, "tests", "test_oracle_known_answers.py"),
                          os.path.join(ROOT, "tests", "test_oracle_traces.py")],
                         env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert "passed" in out.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
