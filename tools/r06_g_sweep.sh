#!/bin/bash
# children kernel at small parent counts under forced tiles-per-item / inline-partial settings (plan_items' cost model check)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1.txt; : > $O
cd $R
for v in "-" "RGL_FUSED_G=1" "RGL_FUSED_G=2" "RGL_FUSED_G=3" "RGL_FUSED_G=6" "RGL_FUSED_G=1 RGL_FUSED_INLINE_PARTIAL=1" "RGL_FUSED_G=2 RGL_FUSED_INLINE_PARTIAL=1" "RGL_FUSED_G=3 RGL_FUSED_INLINE_PARTIAL=1" "RGL_FUSED_G=2 RGL_FUSED_INLINE_PARTIAL=0" "RGL_FUSED_G=5 RGL_FUSED_INLINE_PARTIAL=0"; do
  [ "$v" = "-" ] && vv="" || vv="$v"
  echo "== ${vv:-plan}" >> $O
  env $vv python tools/kiter.py --contraction bf16x6 --quick --reps 150 --parents 256 512 1024 2>&1 | grep "^pair" >> $O
done
cat $O
