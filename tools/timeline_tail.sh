#!/bin/bash
# kernel + copy timeline (with gaps) of the LAST <count> GPU operations of a command: tools/timeline_tail.sh <count> <command...>
# -> gpurun_out/timeline_tail.txt
CNT=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tlt
( cd $GRAFT_REPO_ROOT && timeout 900 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tlt -o t --output-format csv -- "$@" > /tmp/tlt.log 2>&1 )
python - $CNT > $GRAFT_REPO_ROOT/gpurun_out/timeline_tail.txt <<'PY'
import csv, glob, sys
cnt = int(sys.argv[1])
ev = []
for f in glob.glob("/tmp/tlt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob("/tmp/tlt/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
ev = ev[-cnt:]
t0 = ev[0][0]; prev = t0
for s, e, n in ev:
    print("%9.1f us  +%7.1f gap  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
PY
cat $GRAFT_REPO_ROOT/gpurun_out/timeline_tail.txt
