#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_search.py tests/test_gpu_mining.py -x -q -k "long_lists or mining or dense_matches or fused_matches" > $OUT/pytest_big2.txt 2>&1; tail -5 $OUT/pytest_big2.txt
timeout 600 python tools/mining_bench.py --no-whole-call --ks 1000 > $OUT/mining_bench_big2.json 2> $OUT/mining_bench_big2.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06/mining_bench_big2.json'))
for k,v in d['searches'].items():
    for m in ('exact_scores','ids_only'):
        if m in v: print(k, m, 'device_ms %.2f score_kernel_ms %.2f' % (v[m]['device_ms'], v[m]['score_kernel_ms']))
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/mk && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mk -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/mining_one.py t2i 1000 ids 3 > /tmp/mk.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/mk/**/*kernel_stats.csv", recursive=True)[0]
for r in sorted(csv.DictReader(open(f)), key=lambda r: -float(r["TotalDurationNs"]))[:5]:
    print("%-60s calls %6.1f avg_us %9.1f ms/search %8.3f" % (r["Name"][:60], int(r["Calls"]) / 3, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 3e6))
PY
