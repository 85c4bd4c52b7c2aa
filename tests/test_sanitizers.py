"""ASan + UBSan run (CPU only -- GPU sanitizers are not available on this pool) of the oracle and of the host build of
the kernels' SWAR header over 60,000 random / degenerate boards; also cross-checks the two implementations."""
import os
import subprocess

from conftest import REPO


def test_asan_ubsan_clean():
    d = os.path.join(REPO, "tests", "san")
    subprocess.check_call(["make", "-C", d, "-s"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([os.path.join(d, "san_driver")], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches 0" in out.stdout
