#!/bin/bash
# round 5: new dense kernel (16x16x32, 256 x 128 / 256 x 256 tiles) — parity tests, then the S2 shapes and the headline with each tile forced
T=${1:-r05b}; O=$GRAFT_REPO_ROOT/gpurun_out/$T; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_search.py tests/test_gpu_configs.py tests/test_gpu_c_abi.py tests/test_gpu_mining.py -m gpu -x -q > $O/pytest_dense.log 2>&1; echo "pytest rc=$?" >> $O/pytest_dense.log
tail -3 $O/pytest_dense.log
export LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so
for r in 1 2; do
for cb in 4 8 0; do
  for w in flickr coco; do
    LDOT_DEBUG_DENSE_CB=$cb timeout 300 python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r cb=$cb $w: ms/step %.4f  t2i %.4f i2t %.4f kernel_ms %.4f frac %.3f' % (d['ms_per_step'], d['ms_text_to_image'], d['ms_image_to_text'], r['kernel_ms_per_step'], r['frac']), flush=True)
" | tee -a $O/dense_variants.txt
  done
done
done
unset LDOT_DEBUG_DENSE_CB
# headline with the warm-up chunk on each tile (kernel_ms includes the warm-up launch)
for r in 1 2; do for cb in 4 8; do
LDOT_DEBUG_DENSE_CB=$cb timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('round $r cb=$cb headline: ms/step %.3f kernel_ms %.3f frac %.3f' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac']), flush=True)
" | tee -a $O/dense_variants.txt
done; done
# ladder of the fused kernel on this box: product, K rotation, tau = +inf, both, no filter
bash tools/ab.sh "0 32 16 48 17" 2 10 "--no-secondary" | tee $O/ab_ladder.txt
timeout 300 tools/bin/mfma_ceiling 12 2>&1 | head -12 > $O/mfma_ceiling_head.txt
cat $O/mfma_ceiling_head.txt
