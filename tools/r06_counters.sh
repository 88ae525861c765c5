#!/bin/bash
# which counters separate Infinity-Cache hits from HBM reads?  (1) what this rocprofv3 exposes, (2) one PMC pass per candidate on the
# headline command, summed over the score_filter launches of one pass
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > /tmp/avail.txt 2>&1 || rocprofv3 -L > /tmp/avail.txt 2>&1
grep -i -o "\b[A-Z0-9_]*\(EA0\|EA_\|MALL\|DRAM\|HBM\|UMC\|GMI\|FETCH\|WRITE_SIZE\|TCC_MISS\|TCC_HIT\|TCC_REQ\)[A-Za-z0-9_\[\]]*" /tmp/avail.txt | sort -u > $OUT/counters_avail.txt
wc -l $OUT/counters_avail.txt; head -100 $OUT/counters_avail.txt | tr '\n' ' '
rm -f $OUT/pmc_dram.txt
for g in "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM TCC_EA0_RDREQ" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum" "TCC_BUBBLE_sum" "TCC_EA0_RDREQ_GMI_sum TCC_EA0_RDREQ_IO_sum" "TCC_MISS_sum TCC_HIT_sum" "FETCH_SIZE"; do
  rm -rf /tmp/pd
  timeout 600 rocprofv3 --pmc $g --kernel-trace -d /tmp/pd -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pd.log 2>&1
  f=$(find /tmp/pd -name "*counter_collection.csv" | head -1)
  echo "## --pmc $g" >> $OUT/pmc_dram.txt
  if [ -z "$f" ]; then grep -i "error\|invalid\|not" /tmp/pd.log | head -3 >> $OUT/pmc_dram.txt; continue; fi
  python - "$f" >> $OUT/pmc_dram.txt <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'score_filter' in k or 'rescore' in k:
        for c, v in agg[k].items():
            print(f'{k:42s} {c:28s} sum={v:.6g} dispatches={cnt[(k,c)]}')
PY
done
cat $OUT/pmc_dram.txt
