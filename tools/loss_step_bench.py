#!/usr/bin/env python3
"""bench.py's secondary.loss_step on its own (end-to-end and device microseconds of the bidirectional train-step loss, the torch
formulation beside it), plus the single-direction calc at the shapes of tools/loss_bench.py."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
print(json.dumps(bench.loss_step_metrics(torch.device('cuda'), 768), indent=1))
