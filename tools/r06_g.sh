#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_search.py -x -q -k "long_lists" > $OUT/pytest_big3.txt 2>&1; tail -3 $OUT/pytest_big3.txt
cd /tmp && export TMPDIR=/tmp
for d in t2i i2t; do
rm -rf /tmp/mk && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mk -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/mining_one.py $d 1000 ids 3 > /tmp/mk.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/mk/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:6]:
    print("%-60s calls %6.1f avg_us %9.1f ms/search %8.3f" % (r["Name"][:60], int(r["Calls"]) / 3, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 3e6))
PY
done
