#!/usr/bin/env python3
"""bench.py's `secondary.mining_flickr_train` on its own (the mining searches of dvl/hn.py:45-66 at the Flickr30k train set's size, top-50
and top-1000, default and ids-only mode, + the whole sampled_hard_negatives call), for rocprofv3 runs and host profiles:

    python tools/mining_bench.py [--cprofile] [--no-whole-call] [--ks 50,1000]
"""
import argparse, cProfile, io, json, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

ap = argparse.ArgumentParser()
ap.add_argument('--cprofile', action='store_true', help='host profile of one whole sampled_hard_negatives call')
ap.add_argument('--no-whole-call', action='store_true')
ap.add_argument('--ks', default='50,1000')
a = ap.parse_args()
dev = torch.device('cuda:0')
ks = tuple(int(v) for v in a.ks.split(','))
if a.cprofile:
    pr = cProfile.Profile()
    bench.mining_metrics(dev, 768, whole_call=True, ks=())          # warm
    pr.enable()
    out = bench.mining_metrics(dev, 768, whole_call=True, ks=())
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(40)
    print(s.getvalue())
else:
    out = bench.mining_metrics(dev, 768, whole_call=not a.no_whole_call, ks=ks)
print(json.dumps(out, indent=1))
