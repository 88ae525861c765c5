#!/bin/bash
# Which HIP call sits under the 10-80 ms host stalls of a retrieval evaluation?  tools/stall_probe.py under a rocprofv3
# HIP-API + kernel trace; every slow evaluation is reported with the API calls and the kernels inside its window.
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06; cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/st
STALL_PROBE_STAMPS=1 timeout 900 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d /tmp/st -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/stall_probe.py "$@" > /tmp/st.log 2>&1
grep -v amdgpu.ids /tmp/st.log | tail -40
python - <<'PY' > $GRAFT_REPO_ROOT/gpurun_out/r06/stall_trace.txt
import csv, glob, re
api, ker = [], []
for f in glob.glob("/tmp/st/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        api.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"], r.get("Thread_Id", "")))
for f in glob.glob("/tmp/st/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50]))
for f in glob.glob("/tmp/st/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ker.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
api.sort(); ker.sort()
print("api calls", len(api), "kernels/copies", len(ker))
if api:
    print("trace window", api[0][0], api[-1][1])
# the long API calls of the whole run
long_calls = sorted([a for a in api if a[1] - a[0] > 3_000_000], key=lambda a: a[0])
print("\n== API calls longer than 3 ms:")
for s, e, fn, th in long_calls:
    inside = [k for k in ker if k[1] > s and k[0] < e]
    busy = sum(min(k[1], e) - max(k[0], s) for k in inside)
    print("%s  %.2f ms  thread %s  at +%.1f ms  | device busy %.2f ms in %d kernels/copies: %s" % (
        fn, (e - s) / 1e6, th, (s - api[0][0]) / 1e6, busy / 1e6, len(inside), ", ".join("%s %.2fms" % (k[2][:28], (k[1] - k[0]) / 1e6) for k in inside[:6])))
# the probe's own stamps: slow evaluations with their windows in three clocks
for line in open("/tmp/st.log"):
    m = re.match(r"SLOW (\d+) (\d+) mono (\d+) (\d+) boot (\d+) (\d+) real (\d+) (\d+)", line)
    if not m or not api:
        continue
    v = list(map(int, m.groups()))
    for name, s, e in (("mono", v[2], v[3]), ("boot", v[4], v[5]), ("real", v[6], v[7])):
        if api[0][0] <= s <= api[-1][1]:
            print("\n== slow evaluation (n_img %d, index %d): %.2f ms, clock %s" % (v[0], v[1], (e - s) / 1e6, name))
            for a in api:
                if a[1] > s and a[0] < e and a[1] - a[0] > 200_000:
                    print("   api  %-34s %8.3f ms  (+%.3f)" % (a[2], (a[1] - a[0]) / 1e6, (a[0] - s) / 1e6))
            for k in ker:
                if k[1] > s and k[0] < e:
                    print("   dev  %-50s %8.3f ms  (+%.3f)" % (k[2], (k[1] - k[0]) / 1e6, (k[0] - s) / 1e6))
PY
cat $GRAFT_REPO_ROOT/gpurun_out/r06/stall_trace.txt | cut -c1-400 | head -120
