#!/usr/bin/env python3
"""Timing of the retrieval step at the shapes of BASELINE.json configs 1-3 (synthetic stand-ins, SURVEY 8d S2: the real
checkpoints / LMDBs are not available): Flickr30k 1k-test (1 000 images x 5 000 captions) and MSCOCO 5k-test
(5 000 x 25 000), both directions with the reference's un-deduplicated image queries (dvl/trainer.py:160-170), top-100,
index build (add) and search timed separately; results checked against the planted pairs."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lightningdot_amd.indexer import FlatIPIndex

def run(name, n_img, cpi, d=768, k=100):
    torch.manual_seed(7)
    img = torch.randn(n_img, d, device='cuda')
    cap = img.repeat_interleave(cpi, 0) + 0.9 * torch.randn(n_img * cpi, d, device='cuda')
    out = {}
    for direction, x, q, gt in (('txt2img', img, cap, torch.arange(n_img, device='cuda').repeat_interleave(cpi)),
                                ('img2txt', cap, img.repeat_interleave(cpi, 0), None)):
        ix = FlatIPIndex(d)
        if os.environ.get('MODE'): ix.set_option(1, int(os.environ['MODE']))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ix.add(x); torch.cuda.synchronize(); t_add = time.perf_counter() - t0
        ix.search_tensors(q, k); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): s, l = ix.search_tensors(q, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        if gt is not None:
            r1 = float((l[:, 0] == gt).float().mean())
        else:
            r1 = float(((l[:, 0] // cpi) == torch.arange(n_img, device='cuda').repeat_interleave(cpi)).float().mean())
        out[direction] = {'index_rows': x.shape[0], 'queries': q.shape[0], 'add_ms': t_add * 1e3, 'search_ms': dt * 1e3,
                          'queries_per_s': q.shape[0] / dt, 'algorithmic_tflops': 2.0 * q.shape[0] * x.shape[0] * d / dt / 1e12,
                          'recall@1_vs_planted': r1}
    print(json.dumps({name: out}))

run('flickr30k_1k_test_shape', 1000, 5)
run('mscoco_5k_test_shape', 5000, 5)
