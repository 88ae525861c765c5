#!/bin/bash
# where do the waves of the fused kernel wait?  One PMC pass (SQ group) over: the product kernel, its no-filter ablation variant and the
# micro-benchmark's slab loop (tools/mfma_ceiling.hip T16B burst).  SQ_WAIT_ANY = parked at s_waitcnt / s_barrier, SQ_WAIT_INST_ANY = issue
# stall, SQ_ACTIVE_INST_ANY = issuing (quad-cycles, sum ~ SQ_WAVE_CYCLES).   usage (GPU box): tools/pmc_waits.sh > gpurun_out/pmc_waits.txt
G="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
cd /tmp && export TMPDIR=/tmp
agg() { python - "$1" "$2" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
for k in agg:
    if sys.argv[2] in k:
        a = agg[k]; w = a.get('SQ_WAVE_CYCLES', 0) or 1
        print(k)
        for c, v in a.items(): print(f'   {c:28s} {v:.5g}   {v / w * 100:6.2f} % of wave cycles')
        if a.get('GRBM_GUI_ACTIVE'): print('   matrix-pipe busy %.1f %%' % (a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['GRBM_GUI_ACTIVE'] / 8 * 1024) * 100))
PY
}
for v in 0 17 16; do
  rm -rf /tmp/pw; echo "== bench.py, LDOT_DEBUG_VARIANT=$v (0 product, 17 no filter, 16 tau = +inf)"
  LDOT_LIBRARY=$GRAFT_REPO_ROOT/lightningdot_amd/libldot_ablation.so LDOT_DEBUG_VARIANT=$v timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pw -o p --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary > /tmp/pw.log 2>&1
  agg "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" score_filter
done
rm -rf /tmp/pw; echo "== tools/bin/mfma_ceiling 12 t16b"
timeout 300 rocprofv3 --pmc $G --kernel-trace -d /tmp/pw -o p --output-format csv -- $GRAFT_REPO_ROOT/tools/bin/mfma_ceiling 12 t16b > /tmp/pw.log 2>&1
agg "$(find /tmp/pw -name '*counter_collection.csv' | head -1)" ceiling16b
