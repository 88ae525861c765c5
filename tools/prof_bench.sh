#!/bin/bash
# rocprofv3 kernel stats of bench.py (args forwarded)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kb
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kb -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary "$@" > /tmp/kb.log 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/kb/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ldot" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:50], r["Calls"], "avg_us=%.1f"%(float(r["AverageNs"])/1e3), "min=%.1f max=%.1f"%(float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3), "tot_ms=%.3f"%(float(r["TotalDurationNs"])/1e6))
PY
