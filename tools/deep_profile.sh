#!/bin/bash
# rocprofv3 kernel-trace of the value-of-children pair on the configs[4] shape (N = 50, L = 3), f32 and f16 contractions.
#   gpurun -- 'bash tools/deep_profile.sh [parents]'   -> gpurun_out/deep_<variant>/  + per-kernel tables on stdout
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-512}
for v in f32 f16; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/deep_$v -o deep -- python $R/tools/profile_children.py --parents $P --humans 49 --layers 3 --reps 10 --contraction $v > $R/gpurun_out/deep_$v.log 2>&1
  f=$(find $R/gpurun_out/deep_$v -name "*results.db" | head -1)
  echo "== $v"; python $R/tools/rocpd_summary.py $f | head -5
done
