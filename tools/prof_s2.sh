#!/bin/bash
# kernel stats of the S2 evaluation shapes (Flickr-1k / COCO-5k): usage (GPU box) tools/prof_s2.sh <outdir>
O=${1:-$GRAFT_REPO_ROOT/gpurun_out/s2}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in flickr coco; do
  rm -rf /tmp/ks2
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload $w --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_${w}_profiled.json 2> $O/rocprof_$w.err
  cp $(find /tmp/ks2 -name "*kernel_stats.csv" | head -1) $O/${w}_kernel_stats.csv
  echo "== $w"; head -14 $O/${w}_kernel_stats.csv | cut -c1-60,200-330
done
