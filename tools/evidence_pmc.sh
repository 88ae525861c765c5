#!/bin/bash
# the profiler half of tools/evidence.sh on its own: kernel stats of the headline command + PMC passes.  usage: tools/evidence_pmc.sh <tag>
T=${1:-ev}; O=$GRAFT_REPO_ROOT/gpurun_out/ev_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $GRAFT_REPO_ROOT
rm -f $O/pmc.txt
for g in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "TCC_HIT TCC_MISS TCC_REQ" "FETCH_SIZE" "WRITE_SIZE"; do
  bash tools/pmc2.sh "$g" >> $O/pmc.txt 2>&1
done
cat $O/pmc.txt; head -12 $O/kernel_stats.csv | cut -c1-160
