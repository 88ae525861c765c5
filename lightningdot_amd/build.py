"""Build recipe for libldot.so (hand-written HIP for gfx950, C ABI in include/ldot.h).

    python -m lightningdot_amd.build          # (re)build in-tree: lightningdot_amd/libldot.so

hipcc cross-compiles for gfx950 without a GPU, so this runs in the CPU-only build container; the in-tree .so
travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libldot.so')
SOURCES = ['api.hip', 'search.hip', 'scan.hip', 'ivf_api.hip', 'convert.hip', 'score_dense.hip', 'score_narrow.hip', 'score_filter.hip', 'select.hip', 'select_big.hip', 'select_narrow.hip', 'rescore.hip', 'ivf.hip', 'loss.hip']
HEADERS = ['ldot_common.h', 'gemm_ring.h', 'bitonic.h', 'kernels.h', 'pool_walk.h', 'index_state.h', os.path.join('..', '..', 'include', 'ldot.h')]
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, ablation: bool = False) -> str:
    """ablation=True (or `--ablation`): compile the profiling variants of the kernels and their LDOT_DEBUG_* environment hooks
    (-DLDOT_ABLATION) into a SEPARATE library, libldot_ablation.so (selected with LDOT_LIBRARY=...); the product library has no
    such hooks and is never touched by an ablation build."""
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    bdir = os.path.join(PKG, 'build_ablation' if ablation else 'build')
    flags = FLAGS + (['-DLDOT_ABLATION'] if ablation else [])
    lib = os.path.join(PKG, 'libldot_ablation.so') if ablation else LIB
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(bdir, s.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + flags + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for s, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors='replace'))
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f'hipcc failed on {s}\n' + (out.decode(errors='replace') if not verbose else ''))
    if failed:
        raise RuntimeError('libldot.so build failed')
    if force or _stale(lib, objs):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    # a device-side compile error can leave host stubs without their kernels: refuse a library that does not load.
    # (checked in a child process: loading it here would map the system HIP runtime before torch maps its own)
    subprocess.check_call([sys.executable, '-c',
                           'import ctypes, os, sys; ctypes.CDLL(sys.argv[1], mode=os.RTLD_NOW | os.RTLD_LOCAL)', lib])
    return lib


def build_tools(verbose: bool = True) -> str:
    """measurement tools with device code (not part of the product library): tools/bin/mfma_ceiling, sstore_probe, l2_stride_probe, mall_probe"""
    root = os.path.dirname(PKG)
    first = None
    for name, deps in (('mfma_ceiling', [os.path.join(CSRC, 'gemm_ring.h')]), ('sstore_probe', []), ('l2_stride_probe', []), ('mall_probe', [])):
        src = os.path.join(root, 'tools', name + '.hip')
        out = os.path.join(root, 'tools', 'bin', name)
        first = first or out
        if not os.path.exists(src):
            continue
        os.makedirs(os.path.dirname(out), exist_ok=True)
        if _stale(out, [src] + deps):
            cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', CSRC, src, '-o', out]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.check_call(cmd)
    return first


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, ablation='--ablation' in sys.argv))
    if '--tools' in sys.argv:
        print(build_tools())
