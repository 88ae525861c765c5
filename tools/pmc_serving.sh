#!/bin/bash
# HBM-side traffic of the serving workload's index stream (score_narrow_kernel): rocprofv3 PMC, one counter group per pass,
# --kernel-trace only.  Prints per-launch means.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports HALF the bytes of a wide
# coalesced streaming read (MI355X_MICROARCH.md, HBM section) — the doubled value is printed next to the raw one.
cd /tmp && export TMPDIR=/tmp
for g in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT TCC_MISS TCC_REQ"; do
  rm -rf /tmp/pmcs
  timeout 600 rocprofv3 --pmc $g --kernel-trace -d /tmp/pmcs -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload serving --steps 5 --warmup 3 --no-cpu-baseline "$@" > /tmp/pmcs.log 2>&1
  f=$(find /tmp/pmcs -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    key = (r.get('Dispatch_Id'), k)
    if key not in seen:
        seen.add(key); calls[k] += 1
for k in agg:
    if 'score_narrow' in k or 'rescore' in k:
        for c, v in agg[k].items():
            per = v / max(calls[k], 1)
            extra = '  (x2 = %.4g KiB = %.4g GB)' % (2 * per, 2 * per * 1024 / 1e9) if c == 'FETCH_SIZE' else ''
            print(f'{k:42s} {c:12s} launches {calls[k]:3d}  per launch {per:.5g}{extra}')
PY
done
