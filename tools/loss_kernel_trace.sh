#!/bin/bash
# per-kernel GPU durations of the loss path by kernel AND grid size (the shapes of tools/loss_kernels_bench.py differ in grid):
#   tools/loss_kernel_trace.sh > gpurun_out/loss_kernel_trace.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lk
timeout 600 rocprofv3 --kernel-trace -d /tmp/lk -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/loss_kernels_bench.py "$@" > /tmp/lk.log 2>&1
grep "forward" /tmp/lk.log
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/lk/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"][:64], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""))
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
print("%-64s %10s %8s %8s %8s %8s" % ("kernel", "grid", "calls", "avg_us", "min_us", "max_us"))
for (name, gx, gy), d in rows[:40]:
    print("%-64s %10s %8d %8.1f %8.1f %8.1f" % (name, f"{gx}x{gy}", len(d), sum(d) / len(d), min(d), max(d)))
PY
