#!/usr/bin/env python3
"""Headline benchmark: end-to-end queries/sec (+ Recall@1/5/10) of exact top-100 retrieval over a synthetic
1M x 768 index with 10k queries (BASELINE.json configs[3]), on 1/2/4/8 MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one full search pass: query embeddings resident in HBM -> final top-100 (scores + row labels) on the
host.  With N > 1 the index is sharded row-wise (rank r holds N_rows/N rows), every rank owns 1/N of the queries and
a step is all-gather(queries) -> local fused search -> all-to-all(partial top-k) -> merge (lightningdot_amd.sharded).
Total work is fixed as N grows ("strong" scaling).  Rank 0 prints ONE JSON line.

Synthetic data (no network: there is no dataset / checkpoint): index rows i.i.d. N(0,1) fp32, NOT normalised
(the reference scores raw inner products), generated per 125 000-row chunk c from seed 1234 + c; queries
q_i = X[(i * 9973) mod N] + 0.5 * eps_i (seed 4321) so that rank-1 is known and Recall@k is meaningful.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
CHUNK = 125_000


def gen_rows(r0: int, r1: int, d: int, device) -> torch.Tensor:
    """Rows [r0, r1) of the synthetic index; chunk c = rows [c*CHUNK, (c+1)*CHUNK) comes from seed 1234 + c."""
    out = []
    c0, c1 = r0 // CHUNK, (r1 - 1) // CHUNK
    for c in range(c0, c1 + 1):
        g = torch.Generator(device='cpu').manual_seed(1234 + c)
        # generated on the host generator for cross-device reproducibility, in slabs to bound host memory
        blk = torch.randn(CHUNK, d, generator=g, dtype=torch.float32)
        a, b = max(r0, c * CHUNK), min(r1, (c + 1) * CHUNK)
        out.append(blk[a - c * CHUNK:b - c * CHUNK].to(device))
    return torch.cat(out, 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rows', type=int, default=1_000_000)
    ap.add_argument('--queries', type=int, default=10_000)
    ap.add_argument('--dim', type=int, default=768)
    ap.add_argument('--k', type=int, default=100)
    ap.add_argument('--mode', default='auto', choices=['auto', 'dense', 'fused'])
    ap.add_argument('--growth', type=int, default=0, help='fused launch growth percent (0 = library default)')
    ap.add_argument('--warm', type=int, default=0, help='dense warm-up rows (0 = library default)')
    ap.add_argument('--force-sharded', action='store_true', help='run the sharded code path even with one rank')
    ap.add_argument('--force-repeat', action='store_true', help='sharded path: every search runs the verdict + repeat path of the pooled scheme (what a failed pooled search costs; measurement aid)')
    ap.add_argument('--split-bf16', action='store_true',
                    help='LDOT_OPT_PRECISION=1: split-bf16 candidate pass (3 MFMA products per element); not the headline')
    ap.add_argument('--no-optimistic', action='store_true', help='LDOT_OPT_OPTIMISTIC = 0: guaranteed thresholds only (measurement aid; not the headline)')
    ap.add_argument('--scan-order', type=int, default=0, help='LDOT_OPT_SCAN_ORDER: 0 auto (the headline), 1 storage order, 2 scrambled (measurement aid)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary workloads measured after the timed region')
    ap.add_argument('--no-kernel-events', action='store_true',
                    help='measurement aid: do not bracket the score kernels with HIP events (the roofline object is then empty); '
                         'shows what the events themselves cost the timed region')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to exercise the N > 1 '
                                                      'bookkeeping on a one-GPU box together with --all-on-device0)')
    ap.add_argument('--all-on-device0', action='store_true', help='debug: every rank uses cuda:0')
    ap.add_argument('--cpu-sample-queries', type=int, default=4096)
    ap.add_argument('--workload', default='synthetic1m', choices=['synthetic1m', 'flickr', 'coco', 'serving'],
                    help='synthetic1m: BASELINE.json configs[3] (the headline); flickr / coco: the retrieval evaluation of configs[1] / '
                         'configs[2] at the SURVEY 8d S2 stand-in shapes (1 000 / 5 000 images x 5 captions, both directions)')
    ap.add_argument('--duplicated-image-queries', action='store_true',
                    help='flickr / coco workloads: search the image vector of EVERY (caption, image) pair like the reference does '
                         '(dvl/trainer.py:138-139,170: 5 identical searches per image) instead of once per image id')
    ap.add_argument('--normalised', action='store_true',
                    help='second series (SURVEY 8d): rows and queries scaled to unit L2 norm (cosine scores)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run
        import socket
        sock = socket.socket()
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
        sock.close()
        os.execv(sys.executable, [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
                                  f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1', '--master-port', str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    if args.all_on_device0:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import DenseFlatIndexer
    L.require_gpu()
    if args.workload == 'serving':
        return main_serving(args, world, rank, dev, sharded)
    if args.workload != 'synthetic1m':
        return main_s2(args, world, rank, dev, sharded)

    N, Q, D, K = args.rows, args.queries, args.dim, args.k
    # ---- build this rank's shard -------------------------------------------------------------------------
    per = (N + world - 1) // world
    lo, hi = min(rank * per, N), min((rank + 1) * per, N)
    x_local = gen_rows(lo, hi, D, dev)
    # planted queries: q_i = X[g_i] + 0.5 eps_i ; every rank fills the rows it owns, then sum over ranks
    gt = (torch.arange(Q, dtype=torch.int64) * 9973) % N
    geps = torch.Generator(device='cpu').manual_seed(4321)
    eps = torch.randn(Q, D, generator=geps, dtype=torch.float32).to(dev)
    q_all = torch.zeros(Q, D, device=dev)
    own = ((gt >= lo) & (gt < hi)).to(dev)
    q_all[own] = x_local[(gt.to(dev)[own] - lo)]
    if sharded:
        dist.all_reduce(q_all)
    q_all += 0.5 * eps
    if args.normalised:
        x_local = torch.nn.functional.normalize(x_local, dim=1)
        q_all = torch.nn.functional.normalize(q_all, dim=1)
    mode = {'auto': L.MODE_AUTO, 'dense': L.MODE_DENSE, 'fused': L.MODE_FUSED}[args.mode]

    if not sharded:
        ix = DenseFlatIndexer(D)
        ix.index.set_option(L.OPT_MODE, mode)
        ix.index.set_option(L.OPT_PROFILE, 0 if args.no_kernel_events else 1)
        if args.split_bf16:
            ix.index.set_option(L.OPT_PRECISION, 1)
        if args.no_optimistic:
            ix.index.set_option(L.OPT_OPTIMISTIC, 0)
        if args.scan_order:
            ix.index.set_option(L.OPT_SCAN_ORDER, args.scan_order)
        if args.growth:
            ix.index.set_option(L.OPT_GROWTH_PCT, args.growth)
        if args.warm:
            ix.index.set_option(L.OPT_WARM_ROWS, args.warm)
        ix.index.add(x_local)
        flat = ix.index
        q_mine = q_all
        host_s = torch.empty((Q, K), dtype=torch.float32).pin_memory()
        host_l = torch.empty((Q, K), dtype=torch.int64).pin_memory()

        def step():
            # device-resident queries in, top-k on the host out (pinned buffers: the library overlaps the result copies with the
            # re-score of the next query chunk and returns when everything has landed)
            return flat.search_into(q_mine, K, host_s, host_l)
    else:
        from lightningdot_amd.sharded import ShardedFlatIndexer
        sh = ShardedFlatIndexer(D, equal_query_counts=(Q % world == 0))   # equal slices: no per-search exchange of the query counts
        sh.local.index.set_option(L.OPT_MODE, mode)
        sh.local.index.set_option(L.OPT_PROFILE, 1)
        sh.profile_phases = True       # device time of every phase of the exchange (events on the search stream)
        sh.force_repeat = bool(args.force_repeat)
        if args.split_bf16:
            sh.local.index.set_option(L.OPT_PRECISION, 1)
        sh.index_local_shard(list(range(lo, hi)), x_local)
        flat = sh.local.index
        qper = (Q + world - 1) // world
        q_mine = q_all[rank * qper:(rank + 1) * qper].contiguous()
        host_s = torch.empty((q_mine.shape[0], K), dtype=torch.float32).pin_memory()
        host_l = torch.empty((q_mine.shape[0], K), dtype=torch.int64).pin_memory()

        def step():
            # the merge kernel stores the final lists straight into the pinned host buffers (as the re-score kernel does on one GPU)
            return sh.search(q_mine, K, out=(host_s, host_l))

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = dict(launches=0.0, kernel_ms=0.0, flops=0.0, bytes=0.0)
    phases = {}
    barrier()
    import gc
    gc.disable()                       # (a full collection is a 50 ms stall in this process; nothing in a step needs one)
    t0 = time.perf_counter()
    t_prev, step_ms = t0, []
    for _ in range(args.steps):
        step()
        t_now = time.perf_counter()      # (a step ends with its results on the host: every step is complete here)
        step_ms.append((t_now - t_prev) * 1e3)
        t_prev = t_now
        p = flat.last_profile()
        for k_ in prof:
            prof[k_] += p[k_]
        if sharded:
            for k_, v in sh.last_phases.items():
                phases[k_] = phases.get(k_, 0.0) + v
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    ranks_seen = 1
    if sharded:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ones = torch.ones(1, device=dev, dtype=torch.int32)
        dist.all_reduce(ones)                      # every rank that took part in the timed collectives counts itself
        ranks_seen = int(ones.item())
        # every phase's SLOWEST rank (the step waits for it), beside rank 0's own times
        pkeys = sorted(phases)
        pt = torch.tensor([phases[k_] for k_ in pkeys], device=dev, dtype=torch.float64)
        dist.all_reduce(pt, op=dist.ReduceOp.MAX)
        phases_max = dict(zip(pkeys, pt.tolist()))
        if ranks_seen != world:                    # a line measured on fewer ranks than asked for is not a line
            if rank == 0:
                print(f'bench.py: {ranks_seen} ranks took part in the timed collectives, --gpus {world} asked for: no result line', file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(2)

    # ---- quality on the last step's results: Recall@1/5/10 against the planted ground truth ----------------
    s_np, l_np = host_s.numpy(), host_l.numpy()
    if not sharded:
        gt_mine = gt.numpy()
    else:
        gt_mine = gt.numpy()[rank * qper:(rank + 1) * qper]
    hits = np.array([(l_np[:, :t] == gt_mine[:, None]).any(axis=1).sum() for t in (1, 5, 10)], dtype=np.float64)
    nq_mine = np.array([float(len(gt_mine))])
    sorted_ok = bool((np.diff(s_np.astype(np.float64), axis=1) <= 0).all())
    if sharded:
        th = torch.tensor(np.concatenate([hits, nq_mine]), device=dev)
        dist.all_reduce(th)
        hits, nq_tot = th[:3].cpu().numpy(), float(th[3].item())
    else:
        nq_tot = float(nq_mine[0])
    recall = {f'recall@{t}': float(h / nq_tot) for t, h in zip((1, 5, 10), hits)}
    stats = flat.last_stats()

    if rank != 0:
        dist.destroy_process_group()
        return

    value = Q * args.steps / dt
    ach = (prof['flops'] / (prof['kernel_ms'] * 1e-3) / 1e12) if prof['kernel_ms'] > 0 else 0.0
    out = {
        'metric': 'queries/sec', 'value': value, 'unit': 'queries/s', 'n_gpus': world, 'ranks_seen': ranks_seen,
        'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'strong',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'synthetic {N} x {D} bf16 index (fp32 master for exact re-score), {Q} queries, '
                               f'top-{K}, ' + ('unit-L2-normalised rows and queries (cosine; SURVEY S1 second series)'
                                               if args.normalised else
                                               'un-normalised inner product (BASELINE.json configs[3] / SURVEY S1)'),
                   'index_rows': N, 'queries': Q, 'dim': D, 'k': K, 'search_mode': args.mode,
                   'candidate_precision': 'split-bf16 (3 products)' if args.split_bf16 else 'bf16',
                   'parallelism': f'row-sharded index x{world}' if world > 1 else 'single GPU'},
        'step_ms': {'median': float(np.median(step_ms)), 'worst': float(max(step_ms))},   # (rank 0's steps; `ms_per_step` is the mean the contract asks for)
        **recall, 'results_sorted': sorted_ok,
        'overflowed_queries': int(stats['overflowed_queries']),
        'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': ach / PEAK_BF16_TFLOPS, 'traffic': None,
                     'kernel': 'score kernels of rank 0 (score_filter_t16_kernel + warm-up score_dense_t16_kernel)',
                     'launches_per_step': prof['launches'] / max(args.steps, 1),
                     'kernel_ms_per_step': prof['kernel_ms'] / max(args.steps, 1),
                     'flops_per_step': prof['flops'] / max(args.steps, 1)},
    }
    if sharded:
        # rank 0's device time per phase and step (the phases follow each other on one stream: they add up to `total`, which is the
        # step minus the host's share — enqueue, the end-of-search synchronisation, Python)
        out['phases_ms_per_step'] = {k_: v / max(args.steps, 1) for k_, v in phases.items()}
        out['phases_ms_per_step_slowest_rank'] = {k_: v / max(args.steps, 1) for k_, v in phases_max.items()}
        if args.force_repeat:
            out['forced_repeat'] = ('every step ran the pooled scheme\'s verdict (all-reduce SUM of the counts + one host read) and then the whole search again on '
                                    'the shard\'s own thresholds: phase `verdict` and the `repeat:` phases are what a failed pooled search adds')
        out['phases_note'] = ('rank 0, HIP events on the search stream; backend %s%s' %
                              (args.backend, '' if args.backend == 'nccl' else ' (collectives bounce through host copies: exchange phases include them)'))
    if args.no_kernel_events:
        out['roofline']['note'] = 'kernel events disabled (--no-kernel-events): no kernel timing in this run'
    # measured-offline HBM traffic of the dominant kernel (rocprofv3 PMC passes, tools/pmc.sh; see profiles/)
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath) and world == 1 and N == 1_000_000 and Q == 10_000 and D == 768:
        try:
            t = json.load(open(tpath))
            out['roofline']['traffic'] = t['hbm_bytes_per_launch']
            out['roofline']['traffic_source'] = 'offline constant: measured with rocprofv3 PMC passes of this command (profiles/traffic.json), not in this run'
            out['roofline']['traffic_note'] = t['note']
            out['roofline']['traffic_hbm_side'] = t.get('hbm_side_bytes_per_pass')   # (None: no counter separates Infinity-Cache hits from HBM reads here)
            out['roofline']['traffic_hbm_side_note'] = t.get('hbm_side_note')
            out['roofline']['power_note'] = ('offline: while this loop runs the socket draws 1344 W of its 1400 W cap at sclk 1.96 GHz of 2.4 (rocm-smi, '
                                             'profiles/r06_power_probe.txt): the score kernel is power-limited, its rate follows the energy per flop')
        except Exception:
            pass
    if not sharded:
        # PCIe-inclusive variant: fp32 queries start on the HOST (numpy in / numpy out, the reference's calling
        # convention) — reported beside `value`, never as `value`
        q_host = q_all.cpu().numpy()
        flat.search(q_host, K)
        t0 = time.perf_counter()
        for _ in range(2):
            flat.search(q_host, K)
        out['pcie_inclusive'] = {'value': Q * 2 / (time.perf_counter() - t0), 'unit': 'queries/s',
                                 'note': 'fp32 queries in pageable host memory -> scores+labels in host memory'}
    # (secondary shapes BEFORE the CPU baseline: its 256 host threads keep spinning for a while and slow the host side of whatever follows —
    # a COCO-shape evaluation read 6.4 ms after it against 2.4 ms before, profiles/r05_mid_bench.json)
    if not sharded and not args.no_secondary:
        try:
            out['secondary'] = secondary_metrics(dev, flat, D, K)
        except Exception as e:                       # the headline line must not depend on the secondary shapes
            out['secondary'] = {'error': f'{type(e).__name__}: {e}'}
    if not args.no_cpu_baseline and not sharded:
        out['cpu_baseline'], out['parity_vs_cpu_fp32'] = cpu_baseline(x_local, q_all, K, args.cpu_sample_queries,
                                                                      s_np, l_np)
        # (the CPU figures at the S2 shapes are part of `bench.py --workload flickr|coco`: each of those lines carries its own)
    print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


def _time_ms(fn, n, warm=3):
    """mean wall time of n back-to-back calls (the cyclic garbage collector paused: a full collection is a 50 ms stall in this process)"""
    import gc
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    was = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    finally:
        if was:
            gc.enable()


def _median_ms(fn, n, warm=3, full=False, sync=True):
    """median wall time of n individually timed calls, each one complete on the device before the next starts.  One-off host stalls
    (profiles/r05_secondary_outliers.txt: 50-70 ms once in a while, with or without anything this library does differently) do not
    enter; a slowdown that lasts does."""
    import gc
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    was = gc.isenabled()
    gc.disable()
    ts = []
    try:
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            if sync:   # (sync=False: fn returns with its results in place — search_into waits for its own stream — and a second, redundant
                torch.cuda.synchronize()   # device synchronisation would add its ~10 us of host time to a 80-us latency)
            ts.append((time.perf_counter() - t0) * 1e3)
    finally:
        if was:
            gc.enable()
    ts.sort()
    if full:
        return {'p50': ts[len(ts) // 2], 'p99': ts[min(len(ts) - 1, int(len(ts) * 0.99))], 'mean': sum(ts) / len(ts), 'worst': ts[-1], 'calls': len(ts)}
    return ts[len(ts) // 2]


def secondary_metrics(dev, flat_main, D, K):
    """The other workloads of BASELINE.json / SURVEY 8d on the same box, measured AFTER the timed region so that the line the driver
    records carries them (each a few tens of milliseconds of GPU work; none of them enters `value`):
      * retrieval evaluation at the Flickr30k-1k and MSCOCO-5k shapes (configs[1] / [2], `--workload flickr|coco` in full): ms per
        evaluation = text->image over all captions + image->text over every image id once, results on the host, ONE wait for both;
      * serving latency (dvl/utils.py:204-211 retrieve_query): 1 and 64 queries over the headline index and over 123 287 rows (the
        reference demo's COCO index), device query -> pinned host results, with the fraction of the 8 TB/s HBM peak the WHOLE search
        reaches (algorithmic bytes = the bf16 index read once);
      * the approximate index (f-4) over 123 287 clustered rows: one query, 32 probed lists, recall@10 against the exact top-10."""
    from lightningdot_amd.indexer import DenseFlatIndexer, FlatIPIndex
    from lightningdot_amd.synthetic import s2_embeddings
    sec = {}
    # ---- S2 shapes -------------------------------------------------------------------------------------------
    for name, n_img in (('flickr_1k', 1000), ('coco_5k', 5000)):
        img, txt = s2_embeddings(n_img, D, 5, seed=7, device=dev)
        ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
        ix_img.add(img)
        ix_txt.add(txt)
        hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (txt.shape[0], n_img)]
        hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (txt.shape[0], n_img)]

        def step():
            # (the first direction does not wait for its results, LDOT_OPT_DEFER_SYNC: the second search's wait covers both — same stream)
            ix_img.search_into(txt, K, hs[0], hl[0], sync=False)
            ix_txt.search_into(img, K, hs[1], hl[1])
        lat = _median_ms(step, 200, warm=6, full=True)   # (the first evaluations after fresh indexes / pinned buffers carry first-use stalls)
        ms = lat['p50']
        gt = torch.arange(txt.shape[0]) // 5
        # the same pair with DEVICE outputs — what harness.eval_model_on_dataloader itself consumes (Recall@k is reduced on the device); the
        # difference to the pinned-host figure is the 12 bytes per result that leave over PCIe from inside the re-score kernel
        lat_dev = _median_ms(lambda: (ix_img.search_tensors(txt, K), ix_txt.search_tensors(img, K)), 100, warm=4, full=True)
        sec[name] = {'ms_per_evaluation': ms, 'ms_per_evaluation_device_outputs': lat_dev['p50'], 'timing': 'median of 200 individually timed evaluations', 'latency_ms': lat, 'ms_mean': lat['mean'], 'ms_worst': lat['worst'],
                     'queries_searched': int(txt.shape[0] + n_img),
                     'queries_per_s': (txt.shape[0] + n_img) / ms * 1e3,
                     'recall_t2i@1': float((hl[0][:, 0] == gt).float().mean()),
                     'recall_i2t@1': float(((hl[1][:, 0] // 5) == torch.arange(n_img)).float().mean())}
        del ix_img, ix_txt
    # ---- serving latency ---------------------------------------------------------------------------------------
    g = torch.Generator(device='cpu').manual_seed(99)
    x_small = torch.randn(123_287, D, generator=g).to(dev)
    small = FlatIPIndex(D)
    small.add(x_small)
    serving = {}
    for label, ix, n in (('headline_index', flat_main, flat_main.ntotal), ('123k', small, 123_287)):
        for nq in (1, 64):
            rows = (torch.arange(nq, dtype=torch.int64) * 7919) % n
            base = x_small[rows.to(dev)] if ix is small else None
            if base is None:       # rows of the headline index: fetched through the library (the bench does not keep the index tensor)
                base = torch.from_numpy(np.concatenate([ix.get_rows(int(r), 1) for r in rows.tolist()])).to(dev)
            q = base + 0.5 * torch.randn(nq, D, generator=g).to(dev)
            hs_ = torch.empty((nq, K), dtype=torch.float32).pin_memory()
            hl_ = torch.empty((nq, K), dtype=torch.int64).pin_memory()
            lat = _median_ms(lambda: ix.search_into(q, K, hs_, hl_), 200, full=True, sync=False)   # (search_into returns when the results are in hs_ / hl_)
            ms = lat['p50']
            serving[f'{nq}q_x_{label}'] = {'rows': int(n), 'ms': ms, 'latency_ms': lat, 'hbm_frac_whole_search': n * D * 2 / (ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                                           'rank1_ok': bool((hl_[:, 0] == rows).all())}
    sec['serving_latency'] = serving
    # ---- a storage order that is NOT a fair sample order (cluster-sorted rows, what an inverted-file store keeps) ---------------------------
    # the same 10 000 x 1M search; the first search finds out (optimistic thresholds fail their check, the flagged queries are redone), the
    # handle switches to the scrambled tile order, the following searches run in it: latency is data- and history-dependent by design
    try:
        n_so, n_cl = flat_main.ntotal, 40
        gso = torch.Generator(device=dev).manual_seed(77)
        cent = torch.randn(n_cl, D, device=dev, generator=gso)
        per_cl = (n_so + n_cl - 1) // n_cl
        xs = (cent.repeat_interleave(per_cl, dim=0)[:n_so] + 0.5 * torch.randn(n_so, D, device=dev, generator=gso))
        pick = (torch.arange(10_000, device=dev) * 9973) % n_so
        qs = xs[pick] + 0.3 * torch.randn(10_000, D, device=dev, generator=gso)
        ix_so = FlatIPIndex(D)
        ix_so.add(xs)
        del xs
        hs_so = torch.empty((10_000, K), dtype=torch.float32).pin_memory()
        hl_so = torch.empty((10_000, K), dtype=torch.int64).pin_memory()
        runs = []
        for it in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ix_so.search_into(qs, K, hs_so, hl_so)
            torch.cuda.synchronize()
            r = ix_so.last_regime()
            runs.append({'ms': (time.perf_counter() - t0) * 1e3, 'thresholds': r['thresholds'], 'scan_order': r['scan_order'],
                         'redone_queries': int(r['redone_queries']), 'rows': r['rows']})
        sec['sorted_clusters_1m'] = {'what': '10 000 queries x 1M x 768 rows stored sorted in 40 clusters (storage order is not exchangeable), top-100, '
                                             'five consecutive searches of one handle: ms + the regime ldot_index_last_regime reports',
                                     'searches': runs, 'rank1_ok': bool((hl_so[:, 0] == pick.cpu()).float().mean() > 0.999)}
        del ix_so
    except Exception as e:   # (never at the expense of the headline line)
        sec['sorted_clusters_1m'] = {'error': repr(e)}
    # ---- approximate index ---------------------------------------------------------------------------------------
    from lightningdot_amd.ivf import DenseIVFFlatIndexer
    gc = torch.Generator(device=dev).manual_seed(0)
    cent = 0.2 * torch.randn(500, D, device=dev, generator=gc)
    xc = cent[torch.randint(0, 500, (123_287,), device=dev, generator=gc)] + 0.5 * torch.randn(123_287, D, device=dev, generator=gc)
    qc = cent[torch.randint(0, 500, (256,), device=dev, generator=gc)] + 0.5 * torch.randn(256, D, device=dev, generator=gc)
    exact = DenseFlatIndexer(D)
    exact.index_tensor(list(range(123_287)), xc)
    _, el = exact.search_knn_tensors(qc, 10)
    t0 = time.perf_counter()
    ivf = DenseIVFFlatIndexer(D, nprobe=32)
    ivf.index_tensor(list(range(123_287)), xc)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t0
    inv = torch.as_tensor(ivf.index_id_to_db_id, device=dev)
    _, l = ivf.search_knn_tensors(qc, 10, 32, exact_when_cheaper=False)
    orig = torch.where(l >= 0, inv[l.clamp_min(0)], l)
    rec = float(sum(len(set(a.tolist()) & set(b.tolist())) for a, b in zip(orig, el)) / (10 * qc.shape[0]))
    q1 = qc[:1].contiguous()
    ms_ivf = _median_ms(lambda: ivf.search_knn_tensors(q1, 10, 32, exact_when_cheaper=False), 50)
    ms_exact = _median_ms(lambda: exact.search_knn_tensors(q1, 10), 50)
    sec['loss_step'] = loss_step_metrics(dev, D)
    sec['ivf_123k'] = {'nlist': int(ivf.nlist), 'nprobe': 32, 'ms_1_query': ms_ivf, 'ms_1_query_exact_flat': ms_exact,
                       'recall@10_vs_exact': rec, 'build_s': build_s,
                       'data': '123 287 rows = 500 overlapping Gaussian clusters (centroid spread 0.2, noise 0.5), 768-d'}
    del exact, ivf, xc, x_small, small
    sec['mining_flickr_train'] = mining_metrics(dev, D)
    return sec


def loss_step_metrics(dev, D):
    """The in-batch contrastive loss of ONE fine-tuning step (BASELINE configs[4]: global batch 512; train_itm.py:195-222 = two
    BiEncoderNllLoss.calc, bi_encoder.py:615-656, averaged) — forward + backward through autograd, fp32, at 512 x 512 (no hard
    negatives, the reference's fine-tuning configs) and 512 x 1536 (two hard negatives per item): end-to-end microseconds per step as
    the caller sees them (host + device, no synchronisation inside the loop), the device time of one step (events around a step issued behind a busy
    stream, so that host gaps do not count), and the plain torch formulation of the reference on the same GPU beside it."""
    import types
    import torch.nn.functional as F
    from lightningdot_amd.loss import train_step_loss
    out = {}
    for name, bs, nh in (('512x512', 512, 0), ('512x1536', 512, 2)):
        n = bs * (1 + nh)
        g = torch.Generator(device=dev).manual_seed(99)
        txt = (0.2 * torch.randn(n, D, device=dev, generator=g)).requires_grad_()
        img = (0.2 * torch.randn(n, D, device=dev, generator=g) + txt.detach() * (torch.arange(n, device=dev) < bs)[:, None]).requires_grad_()
        args = types.SimpleNamespace(caption_score_weight=0.0, num_hard_negatives=nh)
        batch = dict(sample_size=bs, pos_ctx_indices=list(range(bs)), neg_ctx_indices=list(range(bs, n)))
        pos_t = torch.arange(bs, device=dev)
        one = torch.ones((), device=dev)

        def ours():
            loss, _ic, _sc, _ = train_step_loss(args, txt, img, None, batch)
            loss.backward(one)             # (a cached seed gradient, as train_itm.TRAIN passes: loss.backward() alone launches a fill per step)
            txt.grad = img.grad = None     # (in a training step the towers' backward consumes them; accumulating into leaves would add two kernels)

        def ref():        # the reference's formulation (bi_encoder.py:615-656 twice, train_itm.py:195-222) in torch ops
            def nll(q, c):
                s = q @ c.t()
                ls = F.log_softmax(s, dim=1)
                loss = F.nll_loss(ls, pos_t, reduction='mean')
                correct = (ls.max(1)[1] == pos_t).sum()
                return loss, correct, s
            lt, ct, st = nll(img[:bs], txt)
            li, ci, si = nll(txt[:bs], img)
            loss = 0.5 * lt + 0.5 * li
            _sc = st * 0.5 + si * 0.5
            loss.backward(one)
            txt.grad = img.grad = None

        def device_us(fn):
            # device time of one step: events around a step issued behind a long-running kernel, so that host gaps do not count
            fn()
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda._sleep(2_000_000)
                e0.record()
                fn()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 1e3)
            return best
        txt.grad = img.grad = None
        us = _time_ms(ours, 200, warm=20) * 1e3
        us_dev = device_us(ours)
        txt.grad = img.grad = None
        us_ref = _time_ms(ref, 200, warm=20) * 1e3
        us_ref_dev = device_us(ref)
        txt.grad = img.grad = None
        out[name] = {'bs': bs, 'contexts': n, 'dim': D, 'end_to_end_us': us, 'device_us': us_dev, 'torch_end_to_end_us': us_ref,
                     'torch_device_us': us_ref_dev}
    out['what'] = ('train_step_loss forward + backward (both directions of train_itm.py:195-222), fp32, through autograd; end_to_end = wall '
                   'clock per step over 200 back-to-back steps, device = stream time of one step issued behind a busy stream')
    return out


def mining_metrics(dev, D, whole_call=True, ks=(50, 1000)):
    """The hard-negative mining searches of dvl/hn.py:45-66 at the Flickr30k TRAIN set's size (29 000 images x 145 000 captions; the
    largest retrieval of the reference: both directions over the train set, num_tops = min(max(2 nh + 10, 50), 1000), every epoch), at
    num_tops 50 and 1000:
      * t2i 145 000 x 29 000 and i2t 29 000 x 145 000 (one search per distinct image id: what the harness runs, output-identical to the
        reference's dict comprehension), plus the reference's un-deduplicated 145 000 x 145 000 for context (default mode only);
      * per search: device ms of the whole search (events around it), the score kernels' own ms (LDOT_OPT_PROFILE) and their fraction
        of the bf16 MFMA peak, in the default mode (every candidate re-scored, exact scores) and in the ids-only mode the mining uses
        (LDOT_OPT_RESULT_SET: the same top-k set, boundary candidates re-scored only) with the share of the candidates it gathered;
      * the whole sampled_hard_negatives call (fake towers, batches of 4096, nh = 3 -> top-50), wall clock."""
    import types
    from lightningdot_amd import _lib as L
    from lightningdot_amd.hn import sampled_hard_negatives
    from lightningdot_amd.indexer import FlatIPIndex
    from lightningdot_amd.synthetic import s2_embeddings
    n_img, cpi = 29_000, 5
    img, txt = s2_embeddings(n_img, D, cpi, seed=11, device=dev)
    ix_img, ix_txt = FlatIPIndex(D), FlatIPIndex(D)
    ix_img.add(img)
    ix_txt.add(txt)
    out = {'images': n_img, 'captions': n_img * cpi, 'dim': D, 'searches': {}}

    def device_ms(fn, n=3):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return sorted(ts)[len(ts) // 2]

    img_rep = None
    for k in ks:
        for name, ix, q, modes in (('t2i_145k_x_29k', ix_img, txt, (False, True)), ('i2t_29k_x_145k', ix_txt, img, (False, True)),
                                   ('i2t_undeduplicated_145k_x_145k', ix_txt, None, (False,))):
            if q is None:
                if img_rep is None:
                    img_rep = img.repeat_interleave(cpi, 0)
                q = img_rep
            flops = 2.0 * q.shape[0] * ix.ntotal * D
            row = {'queries': int(q.shape[0]), 'rows': int(ix.ntotal), 'k': k, 'tflop': flops / 1e12}
            for ids_only in modes:
                ms = device_ms(lambda: ix.search_tensors(q, k, ids_only=ids_only))
                ix.set_option(L.OPT_PROFILE, 1)
                _, lab = ix.search_tensors(q, k, ids_only=ids_only)
                prof = ix.last_profile()
                ix.set_option(L.OPT_PROFILE, 0)
                m = {'device_ms': ms, 'score_kernel_ms': prof['kernel_ms'], 'score_kernel_frac': flops / (prof['kernel_ms'] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS,
                     'device_ms_over_score_kernel_ms': ms / prof['kernel_ms'], 'regime': ix.last_regime()['path'] + '/' + ix.last_regime()['thresholds']}
                if ids_only:
                    st = ix.last_set_stats()
                    m['candidates_rescored_share'] = st['rescored'] / max(st['candidates'], 1)
                    m['rows_gathered_GB'] = st['rescored'] * D * 4 / 1e9
                else:
                    m['rows_gathered_GB'] = float((lab >= 0).sum().item()) / k * ((k + max(28, k // 4) + 31) // 32 * 32) * D * 4 / 1e9   # k' rows per query
                if name.startswith('t2i'):
                    m['rank1_ok'] = bool((lab[:, 0] == torch.arange(q.shape[0], device=dev) // cpi).all()) if not ids_only else \
                        bool((lab == (torch.arange(q.shape[0], device=dev) // cpi)[:, None]).any(dim=1).all())
                row['ids_only' if ids_only else 'exact_scores'] = m
                del lab
            out['searches'][f'{name}_top{k}'] = row
    del img_rep
    if whole_call:
        class _Towers:
            def eval(self):
                return self

            def __call__(self, b):
                return b['_q'], b['_ctx'], None
        bs = 4096
        names_t = [f't{j}' for j in range(n_img * cpi)]
        names_i = [f'i{j // cpi}' for j in range(n_img * cpi)]
        img_of = torch.arange(n_img * cpi, device=dev) // cpi
        batches = [dict(txt_index=names_t[b0:b0 + bs], img_fname=names_i[b0:b0 + bs], txts={'input_ids': torch.zeros(min(bs, n_img * cpi - b0), 1, dtype=torch.long)},
                        _q=txt[b0:b0 + bs], _ctx=img[img_of[b0:b0 + bs]]) for b0 in range(0, n_img * cpi, bs)]
        img2txt = {f'i{i}': names_t[i * cpi:(i + 1) * cpi] for i in range(n_img)}
        txt2img = dict(zip(names_t, names_i))
        margs = types.SimpleNamespace(hnsw_index=False, vector_size=D, caption_score_weight=0.0, num_hard_negatives=3)
        g = torch.Generator(device=dev).manual_seed(0)
        sampled_hard_negatives([batches[:3]], margs, _Towers(), img2txt, txt2img, generator=g)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hn_txt, hn_img = sampled_hard_negatives([batches], margs, _Towers(), img2txt, txt2img, generator=g)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        assert len(hn_img) == n_img * cpi and len(hn_txt) == n_img
        out['sampled_hard_negatives_wall_s'] = sorted(ts)[1]
        out['sampled_hard_negatives_what'] = ('whole call, fake towers (embeddings in the batches), 36 batches of 4096, nh = 3 -> top-50 both ways, ids-only '
                                              'searches, positives stripped and negatives drawn on the device, the two {id: [ids]} dicts built; median of 3')
    return out


PEAK_HBM_GBPS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured with a float4 copy)


def main_serving(args, world, rank, dev, sharded):
    """The demo's query path (dvl/utils.py:204-211 retrieve_query): a handful of queries (default 1, --queries) against an
    HBM-resident index of --rows x --dim, top-k, query on the device -> results in pinned host memory.  A step = one search; the
    dominant kernel is the index stream of the narrow search (score_narrow_kernel), HBM-bound: roofline in GB/s."""
    assert not sharded, 'the serving workload is a single-GPU latency measurement'
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import FlatIPIndex
    N, D, K = args.rows, args.dim, args.k
    Q = args.queries if args.queries != 10_000 else 1
    x = gen_rows(0, N, D, dev)
    ix = FlatIPIndex(D)
    ix.add(x)
    g = torch.Generator(device='cpu').manual_seed(4321)
    gt = (torch.arange(Q, dtype=torch.int64) * 9973) % N
    q = x[gt.to(dev)] + 0.5 * torch.randn(Q, D, generator=g).to(dev)
    hs = torch.empty((Q, K), dtype=torch.float32).pin_memory()
    hl = torch.empty((Q, K), dtype=torch.int64).pin_memory()
    ix.set_option(L.OPT_PROFILE, 1)
    for _ in range(max(args.warmup, 3)):
        ix.search_into(q, K, hs, hl)
    prof = dict(launches=0.0, kernel_ms=0.0, flops=0.0, bytes=0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ix.search_into(q, K, hs, hl)
        pp = ix.last_profile()
        for k_ in prof:
            prof[k_] += pp[k_]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stream_bytes = float(N) * D * 2                 # algorithmic bytes of one launch: the bf16 index rows, read once
    kms = prof['kernel_ms'] / max(prof['launches'], 1.0)
    ach = stream_bytes / (kms * 1e-3) / 1e9 if kms > 0 else 0.0
    traffic, tsrc = None, None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'traffic.json')) as f:
            tj = json.load(f)
        if N == 1_000_000 and D == 768 and Q == 1 and 'serving_hbm_bytes_per_launch' in tj:
            traffic, tsrc = tj['serving_hbm_bytes_per_launch'], tj.get('serving_note')
    except OSError:
        pass
    out = {
        'metric': 'queries/sec', 'value': Q * args.steps / dt, 'unit': 'queries/s', 'n_gpus': 1, 'ranks_seen': 1, 'steps': args.steps,
        'warmup': max(args.warmup, 3), 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'serving latency (dvl/utils.py:204-211 retrieve_query shape): {Q} query x {N} rows x {D}-d, top-{K}, '
                               f'exact fp32 re-score, results in pinned host memory', 'rows': N, 'queries': Q, 'dim': D, 'k': K,
                   'parallelism': 'single GPU'},
        'rank1_ok': bool((hl[:, 0] == gt).all()),
        'roofline': {'bound': 'hbm', 'achieved': ach, 'peak': PEAK_HBM_GBPS, 'unit': 'GB/s', 'frac': ach / PEAK_HBM_GBPS,
                     'traffic': traffic, 'traffic_source': tsrc, 'kernel': 'score_narrow_kernel (the index stream)',
                     'launches_per_step': prof['launches'] / max(args.steps, 1), 'kernel_ms_per_launch': kms,
                     'algorithmic_bytes_per_launch': stream_bytes,
                     'whole_search_frac_of_peak': stream_bytes / (dt / args.steps) / 1e9 / PEAK_HBM_GBPS},
    }
    if not args.no_cpu_baseline:
        from oracle import oracle_torch as OT
        cores = os.cpu_count() or 1
        qc, xc = q.cpu(), x.cpu()
        trial = {t: OT.timed(qc, xc, K, t, runs=1)[0] for t in (cores, max(1, cores // 2))}
        threads = min(trial, key=trial.get)
        dtc, _, cl = OT.timed(qc, xc, K, threads, runs=5)
        out['cpu_baseline'] = {'value': Q / dtc, 'unit': 'queries/s', 'cores': int(threads), 'kind': 'port',
                               'sample': f'the same {Q}-query search, oracle_torch.search_blocked (torch.matmul + torch.topk, fp32) at '
                                         f'{threads} threads (better of all / half), median of 5 runs after warm-up',
                               'rank1_mismatches_vs_gpu': int((hl.numpy()[:, 0] != cl.numpy()[:, 0]).sum())}
    print(json.dumps(out), flush=True)


def main_s2(args, world, rank, dev, sharded):
    """Retrieval evaluation at the Flickr30k-1k / MSCOCO-5k shapes (BASELINE.json configs[1] / [2]; synthetic stand-in embeddings,
    SURVEY 8d S2).  A step = the two searches of dvl/trainer.py:160-170: every caption against the image index (top-k) and every
    image — queried once per caption like the reference does — against the caption index; embeddings resident in HBM, results on
    the host.  With N > 1 both indexes are row-sharded and every rank owns 1/N of the queries."""
    import torch.distributed as dist
    from lightningdot_amd import _lib as L
    from lightningdot_amd.indexer import DenseFlatIndexer
    from lightningdot_amd.synthetic import s2_embeddings
    n_img = 1000 if args.workload == 'flickr' else 5000
    D, K, cpi = args.dim, args.k, 5
    img, txt = s2_embeddings(n_img, D, cpi, seed=7, device=dev)
    img_q = img.repeat_interleave(cpi, 0)
    nq = txt.shape[0]
    qper = (nq + world - 1) // world
    qs = slice(rank * qper, min((rank + 1) * qper, nq))

    def make(rows):
        if not sharded:
            ix = DenseFlatIndexer(D)
            ix.index.set_option(L.OPT_PROFILE, 1)
            ix.index.add(rows)
            return ix, ix.index
        from lightningdot_amd.sharded import ShardedFlatIndexer
        per = (rows.shape[0] + world - 1) // world
        lo, hi = min(rank * per, rows.shape[0]), min((rank + 1) * per, rows.shape[0])
        sh = ShardedFlatIndexer(D)
        sh.local.index.set_option(L.OPT_PROFILE, 1)
        sh.index_local_shard(list(range(lo, hi)), rows[lo:hi].contiguous())
        return sh, sh.local.index

    ix_img, flat_img = make(img)
    ix_txt, flat_txt = make(txt)
    # image -> text: the reference searches the image vector of every (caption, image) pair and then keeps ONE result per image id
    # (dict comprehension, dvl/trainer.py:171); the harness (lightningdot_amd/harness.py) searches every id once — the default here
    dedup = not args.duplicated_image_queries
    if dedup:
        iper = (n_img + world - 1) // world
        iqs = slice(rank * iper, min((rank + 1) * iper, n_img))
        q_i = img[iqs].contiguous()
    else:
        iqs = qs
        q_i = img_q[qs].contiguous()
    q_t = txt[qs].contiguous()
    n_mine, n_img_mine = q_t.shape[0], q_i.shape[0]
    hs = [torch.empty((n, K), dtype=torch.float32).pin_memory() for n in (n_mine, n_img_mine)]
    hl = [torch.empty((n, K), dtype=torch.int64).pin_memory() for n in (n_mine, n_img_mine)]
    def one(ix, flat, q, s_out, l_out):
        if not sharded:
            flat.search_into(q, K, s_out, l_out)
        else:
            s, l = ix.search(q, K)
            s_out.copy_(s, non_blocking=True)
            l_out.copy_(l, non_blocking=True)
            torch.cuda.current_stream().synchronize()

    def step():
        one(ix_img, flat_img, q_t, hs[0], hl[0])
        one(ix_txt, flat_txt, q_i, hs[1], hl[1])

    def barrier():
        torch.cuda.synchronize()
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    prof = dict(launches=0.0, kernel_ms=0.0, flops=0.0, bytes=0.0)
    prof_dir = [dict(kernel_ms=0.0, flops=0.0), dict(kernel_ms=0.0, flops=0.0)]
    per_dir = [0.0, 0.0]
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ta = time.perf_counter()
        one(ix_img, flat_img, q_t, hs[0], hl[0])
        tb = time.perf_counter()
        p = flat_img.last_profile()
        one(ix_txt, flat_txt, q_i, hs[1], hl[1])
        tc = time.perf_counter()
        per_dir[0] += tb - ta
        per_dir[1] += tc - tb
        for d_, pp in enumerate((p, flat_txt.last_profile())):
            for k_ in prof:
                prof[k_] += pp[k_]
            for k_ in prof_dir[d_]:
                prof_dir[d_][k_] += pp[k_]
    barrier()
    dt = time.perf_counter() - t0
    ranks_seen = 1
    if sharded:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        ones = torch.ones(1, device=dev, dtype=torch.int32)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
    # Recall@1/5/10 against the planted pairs (dvl/trainer.py:173-188 semantics: text hit iff its image is in the top-k; image hit
    # iff ANY of its captions is)
    gq = torch.arange(qs.start, qs.start + n_mine)
    gt_img = (gq // cpi).numpy()
    gi = (torch.arange(iqs.start, iqs.start + n_img_mine) // (1 if dedup else cpi)).numpy()     # image id of every image query
    l_t, l_i = hl[0].numpy(), hl[1].numpy()
    hits = []
    for t in (1, 5, 10):
        hits.append(float((l_t[:, :t] == gt_img[:, None]).any(axis=1).sum()))
        hits.append(float(((l_i[:, :t] // cpi) == gi[:, None]).any(axis=1).sum()))
    th = torch.tensor(hits + [float(n_mine), float(n_img_mine)], dtype=torch.float64, device=dev)
    if sharded:
        dist.all_reduce(th)
    th = th.cpu().numpy()
    if rank != 0:
        dist.destroy_process_group()
        return
    recall = {f'recall_t2i@{t}': th[2 * i] / th[6] for i, t in enumerate((1, 5, 10))}
    recall.update({f'recall_i2t@{t}': th[2 * i + 1] / th[7] for i, t in enumerate((1, 5, 10))})
    n_iq = n_img if dedup else nq
    flops_step = 2.0 * (nq * img.shape[0] + n_iq * txt.shape[0]) * D
    ach = (prof['flops'] / (prof['kernel_ms'] * 1e-3) / 1e12) if prof['kernel_ms'] > 0 else 0.0
    out = {
        'metric': 'queries/sec', 'value': (nq + n_iq) * args.steps / dt, 'unit': 'queries/s', 'n_gpus': world, 'ranks_seen': ranks_seen,
        'value_note': 'queries actually searched per second (config.queries_searched_per_step x steps / time)',
        'reference_stream_equivalent': {'value': 2 * nq * args.steps / dt, 'unit': 'queries/s',
                                        'note': "rate at which the reference's query stream (one text and one image query per caption: "
                                                "2 x captions per step, dvl/trainer.py:138-139,170) is answered; equals `value` only "
                                                "with --duplicated-image-queries"},
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
        'config': {'workload': f'{args.workload} retrieval evaluation shape (SURVEY S2 stand-in for BASELINE.json configs'
                               f'[{1 if args.workload == "flickr" else 2}]): {n_img} images x {cpi} captions, {D}-d, text->image '
                               f'({nq} queries x {img.shape[0]} rows) + image->text ({n_iq} queries x {txt.shape[0]} rows: '
                               + ('every image id searched once, the result the reference keeps per id' if dedup else
                                  'the image vector of every (caption, image) pair, as the reference searches them') +
                               f'), top-{K}, exact fp32 re-score',
                   'images': n_img, 'captions': int(txt.shape[0]), 'dim': D, 'k': K, 'image_queries_deduplicated': dedup,
                   'queries_searched_per_step': int(nq + n_iq), 'reference_query_stream_per_step': int(2 * nq),
                   'parallelism': f'row-sharded indexes x{world}' if world > 1 else 'single GPU'},
        'ms_text_to_image': per_dir[0] / args.steps * 1e3, 'ms_image_to_text': per_dir[1] / args.steps * 1e3,
        **recall,
        'roofline': {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s', 'frac': ach / PEAK_BF16_TFLOPS,
                     'traffic': None, 'kernel': 'score kernels of rank 0 (both searches)',
                     'launches_per_step': prof['launches'] / max(args.steps, 1),
                     'kernel_ms_per_step': prof['kernel_ms'] / max(args.steps, 1),
                     'flops_per_step': flops_step,
                     'achieved_text_to_image': prof_dir[0]['flops'] / max(prof_dir[0]['kernel_ms'], 1e-9) / 1e9,
                     'achieved_image_to_text': prof_dir[1]['flops'] / max(prof_dir[1]['kernel_ms'], 1e-9) / 1e9,
                     'note': 'latency-class problem (a step is 0.05 - 1 TFLOP): the score kernels are a minority of the step, '
                             'the exact fp32 re-score gather (HBM/L2-bound) is the largest part'},
    }
    if not args.no_cpu_baseline and not sharded:
        from oracle import oracle_torch as OT
        cores = os.cpu_count() or 1
        cpu = {}
        for name, qq, xx in (('t2i', txt, img), ('i2t', img if dedup else img_q, txt)):
            qc, xc = qq.cpu(), xx.cpu()
            # (all hardware threads oversubscribe torch's intra-op pool on these boxes: take the better of all / half the threads)
            trial = {t: OT.timed(qc, xc, K, t, runs=1)[0] for t in (cores, max(1, cores // 2))}
            threads = min(trial, key=trial.get)
            dtc, cs, cl = OT.timed(qc, xc, K, threads, runs=5)
            gl = (hl[0] if name == 't2i' else hl[1]).numpy()
            cpu[name] = {'seconds': dtc, 'threads': threads, 'rank1_mismatches_vs_gpu': int((gl[:, 0] != cl.numpy()[:, 0]).sum())}
        out['cpu_baseline'] = {'value': (nq + n_iq) / (cpu['t2i']['seconds'] + cpu['i2t']['seconds']), 'unit': 'queries/s',
                               'cores': int(cores), 'kind': 'port',
                               'sample': 'the whole step (both searches, the same ' + ('de-duplicated' if dedup else 'duplicated') +
                                         ' image queries as the GPU step), oracle_torch.search_blocked (torch.matmul + torch.topk, '
                                         'fp32) at the better of all / half the hardware threads, median of 5 runs after warm-up', 'detail': cpu}
        if dedup:   # the reference-faithful work (5 identical searches per image) on the same host, stated beside it
            qc, xc = img_q.cpu(), txt.cpu()
            dtr, _, _ = OT.timed(qc, xc, K, cpu['i2t']['threads'], runs=3)
            out['cpu_baseline']['reference_duplicated_stream'] = {
                'value': 2 * nq / (cpu['t2i']['seconds'] + dtr), 'unit': 'queries/s',
                'note': 'image -> text with the reference\'s un-deduplicated queries (dvl/trainer.py:138-139,170): 2 x captions searches '
                        'per step — compare with reference_stream_equivalent, not with value', 'seconds_i2t': dtr}
    print(json.dumps(out), flush=True)
    if sharded:
        dist.destroy_process_group()


def _blas_vendor():
    """what the host GEMMs run on (torch's and numpy's BLAS), for the cpu_baseline line"""
    out = []
    try:
        cfg = torch.__config__.show()
        out.append('torch: ' + ', '.join(sorted({w.strip(' ,') for l in cfg.splitlines() for w in l.split() if w.startswith(('BLAS_INFO=', 'LAPACK_INFO=', 'USE_MKL=', 'USE_MKLDNN='))})))
    except Exception:
        pass
    try:
        import numpy.__config__ as nc
        info = nc.show(mode='dicts') if hasattr(nc, 'show') else {}
        b = info.get('Build Dependencies', {}).get('blas', {})
        out.append('numpy: %s %s' % (b.get('name', '?'), b.get('version', '')))
    except Exception:
        pass
    return '; '.join(out)


def cpu_baseline(x_dev, q_dev, k, nsample, gpu_scores, gpu_labels):
    """The reference's CPU scorer is faiss IndexFlatIP (fp32 sgemm + per-query selection on the host cores); faiss is tried first and
    used when it is importable on the box.  Otherwise the stand-in SURVEY 8d prescribes: oracle_torch.search_blocked (torch.matmul
    over 4096-query x 131072-row tiles + torch.topk(sorted)), with the thread count chosen by a sweep (64 / 128 / 256 / all / half of
    the hardware threads, one run each on the first 1024 queries; numpy's sgemm + argpartition restatement runs in the same sweep, so
    that one library's threading behaviour does not understate the host).  The winner then searches the bounded sample — the first
    `nsample` >= 4096 queries (one full 4096-query tile) against the FULL index — three times; `value` is the median.  Plus a
    one-thread figure on 32 queries.  The same sample is the parity check of the GPU results (rank-1 mismatches and max |delta score|
    against the fp32 CPU path)."""
    from oracle import oracle_np as O          # checker / baseline only — never on the product path
    from oracle import oracle_torch as OT
    cores = os.cpu_count() or 1
    x = x_dev.cpu()
    q = q_dev[:nsample].cpu()
    xn, qn = x.numpy(), q.numpy()
    nsweep = min(1024, q.shape[0])

    def torch_at(threads):
        def run(n):
            old = torch.get_num_threads()
            torch.set_num_threads(threads)
            try:
                s_, l_ = OT.search_blocked(q[:n], x, k)
            finally:
                torch.set_num_threads(old)
            return s_.numpy(), l_.numpy()
        return run

    faiss_used = OT.have_faiss()
    if faiss_used:
        import faiss

        def faiss_at(threads):
            def run(n):
                faiss.omp_set_num_threads(threads)
                return OT.faiss_search(qn[:n], xn, k)
            return run
        cands = {f'faiss.IndexFlatIP (the reference scorer), {t} threads': faiss_at(t) for t in sorted({t for t in (64, 128, 256, cores) if 1 <= t <= cores})}
    else:
        cands = {f'oracle_torch.search_blocked (torch.matmul 4096-query tiles + torch.topk), {t} threads': torch_at(t)
                 for t in sorted({t for t in (64, 128, 256, cores, max(1, cores // 2)) if 1 <= t <= cores})}
        cands['oracle_np.search_fast (numpy BLAS sgemm blocks + argpartition)'] = lambda n: O.search_fast(qn[:n], xn, k)
    OT.search_blocked(q[:8], x[:4096], k)
    O.search_fast(qn[:8], xn[:4096], k)
    sweep = {}
    for name, fn in cands.items():
        t0 = time.perf_counter()
        fn(nsweep)
        sweep[name] = time.perf_counter() - t0
    best = min(sweep, key=sweep.get)
    ts, res = [], None
    for _ in range(3):
        t0 = time.perf_counter()
        res = cands[best](q.shape[0])
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[1]
    cs, cl = res
    n1 = min(32, q.shape[0])
    dt1, _, _ = OT.timed(q[:n1], x, k, 1, runs=3)
    flops = 2.0 * q.shape[0] * x.shape[0] * x.shape[1]
    m_thr = __import__('re').search(r', (\d+) threads', best)
    used = int(m_thr.group(1)) if m_thr else int(cores)      # (numpy's BLAS pool: all hardware threads)
    base = {'value': q.shape[0] / dt, 'unit': 'queries/s', 'cores': used, 'hardware_threads': int(cores), 'kind': 'reference' if faiss_used else 'port',
            'scorer': best, 'faiss_importable': bool(faiss_used), 'blas': _blas_vendor(), 'host_tflops_fp32': flops / dt / 1e12,
            'sample': f'first {q.shape[0]} queries (4096-query tiles) x full {x.shape[0]} x {x.shape[1]} fp32 index, top-{k}; scorer and thread count '
                      f'chosen by a sweep on the first {nsweep} queries; median of 3 runs, {dt:.2f} s per run',
            'sweep_s_on_%d_queries' % nsweep: {n: round(t, 3) for n, t in sweep.items()},
            'value_1thread': n1 / dt1, 'sample_1thread': f'first {n1} queries, torch scorer, 1 thread, median of 3 runs, {dt1:.2f} s per run'}
    gs, gl = gpu_scores[:q.shape[0]], gpu_labels[:q.shape[0]]
    scale = float(np.abs(cs).max()) or 1.0
    # positions where the label differs but the two fp32 scores agree to 1e-4 relative are summation-order ties
    diff = gl != cl
    tie = np.abs(gs.astype(np.float64) - cs.astype(np.float64)) <= 1e-4 * scale
    parity = {'queries': int(q.shape[0]), 'rank1_mismatches': int((gl[:, 0] != cl[:, 0]).sum()),
              'topk_label_mismatches': int(diff.sum()), 'topk_label_mismatches_not_ties': int((diff & ~tie).sum()),
              'max_abs_dscore': float(np.abs(gs.astype(np.float64) - cs.astype(np.float64)).max()),
              'score_scale': scale, 'tolerance': '1e-3 on scores, exact rank-1 (BASELINE.json north_star)'}
    return base, parity


if __name__ == '__main__':
    main()
