#!/bin/bash
# A/B of library variants of the children kernel on ONE box (kiter, interleaved twice):
#   gpurun -- 'bash tools/r06_libab.sh <tag> librgl_hip.so librgl_hip_x.so ...'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/${TAG}.txt
: > $O
cd $R
for rep in 1 2; do
  for lib in "$@"; do
    echo "== rep $rep: $lib" >> $O
    RGL_HIP_LIBRARY=$R/relationalgraphlearning_amd/lib/$lib python tools/kiter.py --contraction bf16x6 --quick --reps ${KITER_REPS:-150} --parents ${KITER_PARENTS:-2048 4096} 2>&1 | grep "^pair\|^check" >> $O
  done
done
cat $O
