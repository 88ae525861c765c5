#!/usr/bin/env python3
"""Where do the 30-90 ms host stalls of a sub-millisecond search come from?  profiles/r06_stall_trace.txt showed the host inside
hipStreamSynchronize / hipLaunchKernel for 30-80 ms while the device had finished after 2 ms, and hipDeviceScheduleSpin did not change it.  This
probe separates the suspects with the scheduler's own bookkeeping (/proc/thread-self/schedstat: ns on a CPU, ns runnable-but-waiting) around
every slow call:
  A. torch only, nothing of this library: a 10-us kernel + torch.cuda.synchronize(), 30 000 times;
  B. a Flickr-shape evaluation (two searches, results in pinned memory), waiting (1) in the runtime (hipStreamSynchronize inside the
     library) and (2) NOT in the runtime: both searches enqueued with LDOT_OPT_DEFER_SYNC, the host polls the last result row that the
     re-score kernel stores straight into pinned memory.
A stall with ~0 ns on-CPU = the thread slept (a late wake-up); with the whole stall on-CPU = it was spinning and the completion became visible
late (device / interconnect side); with a large runnable-wait = the CPU was taken away (hypervisor / other threads)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sched():
    a = open('/proc/thread-self/schedstat').read().split()
    return int(a[0]), int(a[1])


def steal():
    return int(open('/proc/stat').readline().split()[8])


def report(name, ts, extra):
    ts = np.asarray(ts)
    med = float(np.median(ts))
    slow = [(i, t) for i, t in enumerate(ts) if t > max(5.0, 10 * med)]
    print(f'{name}: n {len(ts)}  median {med:.4f} ms  p99 {np.percentile(ts, 99):.4f}  worst {ts.max():.2f}  stalls (> 5 ms): {len(slow)}', flush=True)
    for i, t in slow[:12]:
        on, wait, st = extra[i]
        print(f'    call {i}: {t:.2f} ms   on-CPU {on / 1e6:.2f} ms   runnable-but-waiting {wait / 1e6:.2f} ms   /proc/stat steal ticks +{st}')


import argparse, subprocess
ap = argparse.ArgumentParser()
ap.add_argument('--hog', type=int, default=0, help='busy-loop processes started for the duration of the probe (other tenants of the host, emulated)')
ap.add_argument('--evals', type=int, default=6000)
ap.add_argument('--skip-torch-only', action='store_true')
args = ap.parse_args()
hogs = [subprocess.Popen([sys.executable, '-c', 'while True: pass']) for _ in range(args.hog)]
print(f'LDOT_HOST_WAIT={os.environ.get("LDOT_HOST_WAIT", "(default: poll)")}  busy-loop processes: {args.hog}  host threads: {os.cpu_count()}', flush=True)
dev = torch.device('cuda')
x = torch.zeros(1 << 16, device=dev)
for _ in range(100):
    x.add_(1)
torch.cuda.synchronize()
ts, ex = [], []
for i in range(0 if args.skip_torch_only else 30000):
    s0, st0 = sched(), steal()
    t0 = time.perf_counter()
    x.add_(1)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    s1 = sched()
    ts.append((t1 - t0) * 1e3)
    ex.append((s1[0] - s0[0], s1[1] - s0[1], steal() - st0))
if ts:
    report('A. torch-only (tiny kernel + torch.cuda.synchronize)', ts, ex)

from lightningdot_amd.indexer import FlatIPIndex
from lightningdot_amd.synthetic import s2_embeddings
K, D = 100, 768
img, txt = s2_embeddings(1000, D, 5, seed=7, device=dev)
ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
ix_img.add(img)
ix_txt.add(txt)
hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], 1000)]
hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], 1000)]
l0, l1 = hl[0].numpy(), hl[1].numpy()
for mode in ('the library waits (LDOT_HOST_WAIT decides how)', 'no wait in the library (Python polls the pinned results)'):
    ts, ex = [], []
    for i in range(args.evals):
        if mode.startswith('no'):
            l0[:, -1] = -7
            l1[:, -1] = -7
        s0, st0 = sched(), steal()
        t0 = time.perf_counter()
        ix_img.search_into(txt, K, hs[0], hl[0], sync=False)
        if mode.startswith('no'):
            ix_txt.search_into(img, K, hs[1], hl[1], sync=False)
            while l1[-1, -1] == -7 or (l1[:, -1] == -7).any() or (l0[:, -1] == -7).any():
                pass
        else:
            ix_txt.search_into(img, K, hs[1], hl[1])
        t1 = time.perf_counter()
        s1 = sched()
        ts.append((t1 - t0) * 1e3)
        ex.append((s1[0] - s0[0], s1[1] - s0[1], steal() - st0))
    torch.cuda.synchronize()
    report('B. Flickr-shape evaluation, ' + mode, ts, ex)

for h in hogs:
    h.kill()
