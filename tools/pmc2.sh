#!/bin/bash
# usage: tools/pmc2.sh "<counters>"  -> per-kernel sums for one PMC group
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcx
timeout 600 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmcx -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pmcx.log 2>&1
f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-36:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in agg:
    if 'score_filter' in k:
        for c, v in agg[k].items():
            print(f'{k:38s} {c:32s} {v:.5g}')
PY
