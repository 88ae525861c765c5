#!/bin/bash
# round-5 state check in one gpurun call: GPU tests, smoke, headline bench, loss step bench, kernel stats of the bench command
T=${1:-r05a}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
[ -x tools/bin/mfma_ceiling ] && timeout 300 tools/bin/mfma_ceiling 12 > $O/mfma_ceiling.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $GRAFT_REPO_ROOT
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $O/timeline.txt 2>/dev/null
tail -3 $O/pytest_gpu.log; tail -1 $O/smoke.log; cut -c1-1500 $O/bench.json; head -8 $O/kernel_stats.csv | cut -c1-200
