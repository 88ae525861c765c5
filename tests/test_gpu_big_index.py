"""Byte offsets beyond 2^32 (tools/big_index_check.py): a 3M x 768 index — 4.3 GiB of bf16 shadow, 8.6 GiB of fp32 master — searched through
every regime (narrow, 64 queries, fused scan in storage and scrambled tile order, dense) against an fp64 brute-force scan on the GPU.
(8M rows — offsets beyond 2^33 — ran by hand: profiles/r04_big_index_8m.txt.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_three_million_rows_every_search_regime():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'big_index_check.py'), '3000000', '768'], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count('\nok ') + r.stdout.startswith('ok ') == 6 and 'FAIL' not in r.stdout, r.stdout[-3000:]
