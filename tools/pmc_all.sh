#!/bin/bash
# usage: tools/pmc_all.sh "<counters>" [kernel-name filter]  -> per-kernel sums for one PMC group over bench.py (1 step + 1 warm-up + pcie passes)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmca
timeout 600 rocprofv3 --pmc $1 --kernel-trace -d /tmp/pmca -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/pmca.log 2>&1
f=$(find /tmp/pmca -name "*counter_collection.csv" | head -1)
python - "$f" "${2:-ldot}" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    if sys.argv[2] not in r['Kernel_Name']: continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in agg:
    print(k, '  '.join(f'{c}={v:.4g}' for c, v in sorted(agg[k].items())))
PY
