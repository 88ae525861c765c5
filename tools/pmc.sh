#!/bin/bash
# usage: tools/pmc.sh <tag> [extra bench args]   -> gpurun_out/pmc_<tag>.txt : per-kernel sums of PMC counters
# (each --pmc group is its own rocprofv3 run; only --kernel-trace is combined with --pmc)
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PMCG=(
 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
 "SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE"
 "TCC_HIT TCC_MISS TCC_REQ TCC_EA0_RDREQ"
 "FETCH_SIZE"
 "WRITE_SIZE TCP_TCC_READ_REQ_LATENCY TCP_TCC_READ_REQ"
)
rm -f $OUT/pmc_$TAG.txt
i=0
for g in "${PMCG[@]}"; do
  rm -rf /tmp/pmc_$i
  timeout 600 rocprofv3 --pmc $g --kernel-trace -d /tmp/pmc_$i -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary "$@" > /tmp/pmc_$i.log 2>&1
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$f" >> $OUT/pmc_$TAG.txt <<'PY'
import csv, sys, collections
f = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    cnt[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'score' in k or 'select' in k or 'rescore' in k:
        for c, v in agg[k].items():
            print(f'{k:42s} {c:28s} sum={v:.6g} dispatches={cnt[(k,c)]}')
PY
  i=$((i+1))
done
cat $OUT/pmc_$TAG.txt
