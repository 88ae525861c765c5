#!/bin/bash
# kernel timeline of ONE forward + backward of the in-batch loss at the config-5 shapes (HIP path and the torch formulation of the
# reference), from a rocprofv3 kernel trace of tools/loss_bench.py: tools/loss_timeline.sh > gpurun_out/loss_timeline.txt
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/lt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/lt -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/loss_bench.py --once > /tmp/lt.log 2>&1
cat /tmp/lt.log | grep -v amdgpu.ids
python - <<'PY'
import csv, glob
ev = []
for f in glob.glob("/tmp/lt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
ev.sort()
# loss_bench.py --once brackets every measured fwd+bwd with a marker kernel: torch.zeros(7, ...) fills of 7 elements are rare enough;
# simpler: print the LAST 40 kernels before each 'marker' (an elementwise add on a 12345-element tensor)
marks = [i for i, e in enumerate(ev) if "12345" in e[2]]
print("kernels:", len(ev))
PY
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/lt/**/*kernel_stats.csv", recursive=True)[0]
print("---- kernel stats (all shapes, all repetitions) ----")
for r in list(csv.DictReader(open(f)))[:25]:
    print("%-70s calls %6s  avg_us %8.1f  total_ms %8.3f" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
