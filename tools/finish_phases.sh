#!/bin/bash
# duration of narrow_finish_kernel cut short after its n-th phase (ablation library, LDOT_DEBUG_FINISH_PHASE = 1..5, 0 = whole kernel):
# 9 empty launch, 8 run maxima loaded, 1 threshold, 2 + run list, 3 + collect, 4 + top-k' / list, 5 + exact re-score, 0 + final order and output.   usage: finish_phases.sh <rows>
export LDOT_LIBRARY=$GRAFT_REPO_ROOT/lightningdot_amd/libldot_ablation.so
for p in 9 8 1 2 3 4 5 0; do
  LDOT_DEBUG_FINISH_PHASE=$p bash $GRAFT_REPO_ROOT/tools/serving_timeline.sh ${1:-123287} 1 2>/dev/null | grep narrow_finish | sed "s/^/phase $p: /"
done
