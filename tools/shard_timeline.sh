#!/bin/bash
# kernel timeline of the LAST search of tools/shard_timeline.py: tools/shard_timeline.sh <rows> <local|agreed>
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st
timeout 600 rocprofv3 --kernel-trace -d /tmp/st -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/shard_timeline.py "$@" > /tmp/st.log 2>&1
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
ev.sort()
starts = [i for i, e in enumerate(ev) if "convert_rows_kernel" in e[2]]
i0 = starts[-1]
t0 = ev[i0][0]; prev = t0
for s, e, n in ev[i0:]:
    print("%9.1f us  +%7.1f gap  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
PY
