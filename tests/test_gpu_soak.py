"""A short run of tools/soak.py (randomised differential test of every kernel and both API faces against the CPU
oracle) inside the GPU suite; the long runs are recorded in profiles/r01_soak.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [1, 2])
def test_short_soak(seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "6", str(seed)], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "PASSED" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
