#!/bin/bash
# BASELINE configs[4] per-GPU share (N = 50, L = 3, D = 2, w = 2, 256 roots): bench lines + rocprofv3 kernel trace.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in f32 f16; do
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5_$v -o c5 -- python $R/bench.py --roots 256 --humans 49 --layers 3 --steps 50 --warmup 10 --cpu-seconds 0 --contraction $v > $R/gpurun_out/c5_$v.log 2>&1
  f=$(find $R/gpurun_out/c5_$v -name "*results.db" | head -1)
  echo "== $v"; grep "^{" $R/gpurun_out/c5_$v.log; python $R/tools/rocpd_summary.py $f > $R/gpurun_out/c5_$v.md; head -12 $R/gpurun_out/c5_$v.md
done
