#!/bin/bash
# rocprofv3 kernel stats of the single-query serving shape (tools/smallq1.py)
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sq
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/sq -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/smallq1.py > /dev/null 2>&1
python - <<PY
import csv,glob
f=glob.glob("/tmp/sq/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ldot" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:50], r["Calls"], "avg_us=%.1f"%(float(r["AverageNs"])/1e3), "min=%.1f max=%.1f"%(float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3), "tot_ms=%.3f"%(float(r["TotalDurationNs"])/1e6))
PY
