"""The ctypes binding printed in INTEGRATION.md (Option B, the stub a reference maintainer would add next to
dvl/indexer/faiss_indexers.py) is executed as written against libldot.so and checked against the oracle."""
import os
import re

import numpy as np
import pytest

from oracle import oracle_np as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_integration_md_ctypes_stub_runs():
    from lightningdot_amd import _lib as L
    L.require_gpu()
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    stub = [b for b in blocks if 'class IndexFlatIP' in b]
    assert len(stub) == 1
    code = stub[0].replace("ctypes.CDLL('libldot.so')", "ctypes.CDLL(%r)" % os.path.join(ROOT, 'lightningdot_amd', 'libldot.so'))
    ns = {}
    exec(compile(code, 'INTEGRATION.md', 'exec'), ns)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3000, 96)).astype(np.float32)
    q = rng.standard_normal((17, 96)).astype(np.float32)
    ix = ns['IndexFlatIP'](96)
    ix.add(x)
    assert ix.ntotal == 3000
    s, l = ix.search(q, 10)
    ref = O.FlatIP(96)
    ref.add(x)
    so, lo = ref.search(q, 10)
    np.testing.assert_array_equal(l, lo)
    np.testing.assert_allclose(s, so, rtol=0, atol=1e-4)
