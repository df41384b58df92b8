"""Randomised parity of the fused lift+Gram kernels against the numpy oracle (a fixed-seed slice of tools/gram_fuzz.py, which round 6
ran over 21 000 cases after rebuilding the Kronecker kernel's lift): model types x nzeta x m x degree x dictionary kinds x snapshot
counts around every tile boundary (1 .. 25, 511 .. 513, 5 999 / 6 001: the prelift threshold).  G symmetric bit for bit, repeatable
bit for bit, 2e-12 of the oracle's Grams."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_dictionaries_and_snapshot_counts_against_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gram_fuzz.py"), "160", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "cases OK" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
    done = int(r.stdout.strip().splitlines()[-1].split()[0])
    assert done >= 100, r.stdout[-300:]
