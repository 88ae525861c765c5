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


@pytest.mark.gpu
def test_randomised_sharded_exchange_sweep():
    """tools/fuzz_sharded.py: 2..8 shards of random sizes (empty and tiny ones included) played by one process, statistics reduced
    like the collectives would; pooled statistics proven by the count check or repeated on own thresholds; merged lists bit-identical
    to the plain search of the whole index.  Rows that are exchangeable between the shards must never need the repeat."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'fuzz_sharded.py'), '40', '5'], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert '40/40 passed' in r.stdout
    bad = [ln for ln in r.stdout.splitlines() if 'repeated=True' in ln and 'order=3' not in ln]
    assert not bad, bad
