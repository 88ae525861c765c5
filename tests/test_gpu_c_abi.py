"""A pure-C client of include/ldot.h (tests/c/abi_smoke.c: no Python, no torch in the process) is compiled with gcc,
linked against libldot.so and run: the drop-in boundary is a plain C ABI."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_pure_c_client(tmp_path):
    libdir = os.path.join(ROOT, 'lightningdot_amd')
    assert os.path.exists(os.path.join(libdir, 'libldot.so')), 'run __graft_entry__.build() first'
    exe = str(tmp_path / 'abi_smoke')
    cc = shutil.which('gcc') or shutil.which('cc')
    assert cc, 'no C compiler'
    subprocess.run([cc, '-O2', '-I', os.path.join(ROOT, 'include'), os.path.join(ROOT, 'tests', 'c', 'abi_smoke.c'), '-o', exe,
                    '-L', libdir, '-lldot', '-Wl,-rpath,' + libdir, '-Wl,-rpath,/opt/rocm/lib', '-lm'], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'abi_smoke ok' in r.stdout
