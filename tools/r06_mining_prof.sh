#!/bin/bash
# rocprofv3 kernel stats of the four ids-only mining searches (+ the exact-score ones at top-50) -> gpurun_out/r06/mining_kernels_*.txt
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/r06; cd /tmp && export TMPDIR=/tmp
for cfg in "t2i 50 ids" "i2t 50 ids" "t2i 1000 ids" "i2t 1000 ids" "t2i 50 exact" "t2i 1000 exact"; do
  set -- $cfg
  rm -rf /tmp/mk
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/mk -o k --output-format csv -- python $GRAFT_REPO_ROOT/tools/mining_one.py $1 $2 $3 3 > /tmp/mk.log 2>&1
  out=$GRAFT_REPO_ROOT/gpurun_out/r06/mining_kernels_$1_top$2_$3.txt
  grep -v amdgpu.ids /tmp/mk.log | tail -2 > $out
  python - >> $out <<'PY'
import csv, glob
f = glob.glob("/tmp/mk/**/*kernel_stats.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
tot = sum(float(r["TotalDurationNs"]) for r in rows if "ldot" in r["Name"] or "rocclr" in r["Name"])
print("# per SEARCH (3 searches in the run; add / synthetic-data kernels excluded from the total): name, calls per search, avg us, total ms per search")
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    if "ldot" in r["Name"] or "rocclr" in r["Name"]:
        print("%-70s calls %6.1f  avg_us %9.1f  ms/search %8.3f" % (r["Name"][:70], int(r["Calls"]) / 3, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 3e6))
print("total ldot kernels per search: %.3f ms" % (tot / 3e6))
PY
  cat $out | cut -c1-200
done
