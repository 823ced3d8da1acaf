#!/bin/bash
# A/B of environment variants of the children kernel on ONE box: kiter at two parent counts per variant, interleaved twice.
#   gpurun -- 'bash tools/r06_ab.sh <tag> "VAR=1" "VAR=2 OTHER=3" ...'   ("-" = no variables)
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/${TAG}.txt
: > $O
cd $R
for rep in 1 2; do
  for v in "$@"; do
    [ "$v" = "-" ] && vv="" || vv="$v"
    echo "== rep $rep: ${vv:-baseline}" >> $O
    env $vv python tools/kiter.py --contraction bf16x6 --quick --reps ${KITER_REPS:-150} --parents ${KITER_PARENTS:-2048 4096} 2>&1 | grep "^pair\|^check" >> $O
  done
done
cat $O
