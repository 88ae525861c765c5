"""Bounded run of the randomised parity sweep (tools/fuzz_search.py): random (n, nq, d, k, mode, normalise, clustered
data, incremental add) around the thresholds of the search orchestration, each checked against the fp64 truth."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_randomised_parity_sweep():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_search.py'), '16', '11'], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '16/16 passed' in r.stdout
