#!/bin/bash
# rocprofv3 kernel stats of an arbitrary python command: tools/prof_cmd.sh <script> [args]
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kc
S=$1; shift
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kc -o k --output-format csv -- python $GRAFT_REPO_ROOT/$S "$@" > /tmp/kc.log 2>&1
tail -3 /tmp/kc.log | cut -c1-1500
python - <<PY
import csv,glob
f=glob.glob("/tmp/kc/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "ldot" in r["Name"] or "rocclr" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us=%.1f"%(float(r["AverageNs"])/1e3), "min=%.1f max=%.1f"%(float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3), "tot_ms=%.3f"%(float(r["TotalDurationNs"])/1e6))
PY
