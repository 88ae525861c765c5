O=$GRAFT_REPO_ROOT/gpurun_out/ev_r05_chunks; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
cd $GRAFT_REPO_ROOT
for g in "TCC_HIT TCC_MISS TCC_REQ" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  bash tools/pmc2.sh "$g" >> $O/pmc.txt 2>&1
done
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $O/timeline.txt 2>/dev/null
timeout 300 python tools/shard_floor.py 2>&1 | grep -E "^\(|merged|cannot" > $O/shard_floor.txt
head -4 $O/kernel_stats.csv | cut -c1-200; cat $O/pmc.txt; cat $O/shard_floor.txt; cut -c1-110 $O/timeline.txt | head -30
