#!/bin/bash
# Round evidence in ONE gpurun call: GPU tests, smoke, headline bench (+ CPU baselines), S2-shape benches, serving latency, MFMA ceiling
# microbenchmark, rocprofv3 kernel stats of the bench command, PMC passes (separate --pmc runs, --kernel-trace only), per-pass timeline,
# 2-rank dry run on one GPU, vendor GEMM reference, single-query kernel timelines (exact and inverted-file search).
# usage (GPU box): tools/evidence.sh <tag>   -> gpurun_out/ev_<tag>/
T=${1:-ev}; O=$GRAFT_REPO_ROOT/gpurun_out/ev_$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke(); print("smoke ok")' > $O/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --normalised --no-cpu-baseline > $O/bench_normalised.json 2>> $O/bench.err
timeout 600 python bench.py --workload flickr --steps 50 --warmup 5 > $O/bench_flickr.json 2>> $O/bench.err
timeout 600 python bench.py --workload coco --steps 20 --warmup 3 > $O/bench_coco.json 2>> $O/bench.err
timeout 600 python bench.py --gpus 2 --all-on-device0 --backend gloo --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_2ranks_gloo_one_gpu.json 2>> $O/bench.err
timeout 300 python tools/serving_latency.py > $O/serving_latency.jsonl 2>> $O/bench.err
timeout 600 python bench.py --workload serving --steps 50 --warmup 5 > $O/bench_serving.json 2>> $O/bench.err
timeout 600 python bench.py --workload serving --rows 123287 --steps 50 --warmup 5 > $O/bench_serving_123287.json 2>> $O/bench.err
[ -x tools/bin/mfma_ceiling ] && timeout 300 tools/bin/mfma_ceiling 12 > $O/mfma_ceiling.txt 2>&1
timeout 300 python tools/gemm_ref.py > $O/vendor_gemm.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_profiled.json 2> $O/rocprof.err
cp $(find /tmp/kt -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rm -rf /tmp/kts
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kts -o k --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --workload serving --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_serving_profiled.json 2>> $O/rocprof.err
cp $(find /tmp/kts -name "*kernel_stats.csv" | head -1) $O/serving_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for g in "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "TCC_HIT TCC_MISS TCC_REQ" "FETCH_SIZE" "WRITE_SIZE"; do
  bash tools/pmc2.sh "$g" >> $O/pmc.txt 2>&1
done
bash tools/timeline.sh > /dev/null 2>&1; cp gpurun_out/timeline.txt $O/timeline.txt 2>/dev/null
bash tools/pmc_serving.sh > $O/pmc_serving.txt 2>&1
bash tools/serving_timeline.sh 1000000 > $O/serving_timeline_1m.txt 2>&1
bash tools/serving_timeline.sh 123287 > $O/serving_timeline_123287.txt 2>&1
timeout 600 python tools/ivf_bench.py > $O/ivf_bench.jsonl 2>> $O/bench.err
timeout 600 python tools/ivf_bench.py 1000000 curve > $O/ivf_recall_curve.jsonl 2>> $O/bench.err
timeout 600 python tools/overflow_cases.py > $O/overflow_cases.jsonl 2>> $O/bench.err
timeout 300 python tools/shard_floor.py > $O/shard_floor.txt 2>> $O/bench.err
bash tools/timeline_tail.sh 10 python tools/ivf_one.py > $O/ivf_timeline.txt 2>&1
tail -3 $O/pytest_gpu.log; cat $O/smoke.log | tail -1; cut -c1-600 $O/bench.json; cat $O/pmc.txt
