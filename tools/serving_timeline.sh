#!/bin/bash
# kernel timeline of ONE single-query search (tools/serving_latency.py shape): rocprofv3 kernel trace, last search printed with gaps
# usage: tools/serving_timeline.sh <rows> [queries]   -> gpurun_out/serving_timeline_<rows>.txt
N=${1:-1000000}; NQ=${2:-1}
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/stl
cat > /tmp/stl_run.py <<PY
import os, sys, torch
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
from lightningdot_amd.indexer import FlatIPIndex
torch.manual_seed(0)
x = torch.randn($N, 768, device='cuda')
ix = FlatIPIndex(768); ix.add(x)
q = x[:$NQ] + 0.5 * torch.randn($NQ, 768, device='cuda')
hs = torch.empty(($NQ, 100), dtype=torch.float32).pin_memory(); hl = torch.empty(($NQ, 100), dtype=torch.int64).pin_memory()
for _ in range(8):
    ix.search_into(q, 100, hs, hl); torch.cuda.synchronize()
PY
timeout 600 rocprofv3 --kernel-trace -d /tmp/stl -o t --output-format csv -- python /tmp/stl_run.py > /tmp/stl.log 2>&1
python - > $GRAFT_REPO_ROOT/gpurun_out/serving_timeline_$N.txt <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/stl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
ev.sort()
# a search starts with the index scan (a few fp32 device queries are converted by the scan itself; larger batches by a conversion kernel first)
starts = [i for i, e in enumerate(ev) if "score_narrow" in e[2]]
i0 = starts[-1]
if i0 > 0 and "convert_rows" in ev[i0 - 1][2] and ev[i0][0] - ev[i0 - 1][1] < 20000:
    i0 -= 1
t0 = ev[i0][0]; prev = t0
for s, e, n in ev[i0:]:
    print("%9.1f us  +%7.1f gap  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
print("total %.1f us" % ((prev - t0) / 1e3))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/serving_timeline_$N.txt
