#!/bin/bash
# per-step timeline of bench.py from a rocprofv3 kernel + memory-copy trace: kernels/copies of the LAST search pass with gaps
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary "$@" > /tmp/tl.log 2>&1
python - > $GRAFT_REPO_ROOT/gpurun_out/timeline.txt <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:44]))
for f in glob.glob("/tmp/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
ev.sort()
# the timed steps: find the last 'init_lists' kernel of the device-resident passes (before the pcie-inclusive ones) -> print 1 pass
starts = [i for i, e in enumerate(ev) if "init_lists" in e[2]]
# passes: warmup 2 + steps 3 + pcie 3 = 8; take pass index 4 (last timed)
i0 = starts[4]; i1 = starts[5]
# include the convert kernel that precedes init_lists
while i0 > 0 and "convert_rows" in ev[i0 - 1][2]: i0 -= 1
t0 = ev[i0][0]; prev = t0
for s, e, n in ev[i0:i1]:
    if "convert_rows" in n and s > t0 + 1000000: break
    print("%9.1f us  +%7.1f gap  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n))
    prev = max(prev, e)
PY
