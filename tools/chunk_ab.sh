#!/bin/bash
# row chunks of the fused scan (one launch per 256-tile chunk, cursors carried over, one select): previous library vs this one (product builds),
# then the ablation builds: previous kernel, this kernel with chunks off (LDOT_DEBUG_CHUNK_TILES=0) and with 128 / 256 / 512-tile chunks
bash tools/ab_lib.sh "lightningdot_amd/libldot_prev.so lightningdot_amd/libldot.so" 3 20
run() {
  env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$*: ms/step %.3f kernel_ms/step %.3f frac %.3f recall@1 %.3f overflow %d' % (d['ms_per_step'], r['kernel_ms_per_step'], r['frac'], d['recall@1'], d['overflowed_queries']), flush=True)
"
}
for rep in 1 2; do
run LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation_prev.so
run LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so LDOT_DEBUG_CHUNK_TILES=0
run LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so LDOT_DEBUG_CHUNK_TILES=256
run LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so LDOT_DEBUG_CHUNK_TILES=128
run LDOT_LIBRARY=$PWD/lightningdot_amd/libldot_ablation.so LDOT_DEBUG_CHUNK_TILES=512
done
