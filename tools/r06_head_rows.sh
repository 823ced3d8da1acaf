#!/bin/bash
# The value head's rows in the training step (batch 100): head_rows_kernel (weights straight from L2) against mlp_rows_kernel (weights
# staged in LDS): phase cycles of both (debug build) and their durations in a kernel trace of the captured step, then the step itself.
#   gpurun -- 'bash tools/r06_head_rows.sh' -> gpurun_out/r06_head_rows.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
{
  echo "== phase cycles (tools/head_phases.py 5 100; librgl_hip_timing_head.so)"
  echo "-- head_rows_kernel"; timeout 200 python tools/head_phases.py 5 100 2>&1 | tail -10
  echo "-- mlp_rows_kernel (RGL_HEAD_ROWS_DIRECT=0)"; RGL_HEAD_ROWS_DIRECT=0 timeout 200 python tools/head_phases.py 5 100 2>&1 | tail -10
  echo; echo "== kernel trace of the captured step (tools/r06_head_rows_trace.sh)"
  timeout 600 bash tools/r06_head_rows_trace.sh
  echo; echo "== MPRLTrainer.optimize_batch (tools/trainer_time.py), captured step, index-sampled batches"
  for d in 1 0; do echo "-- RGL_HEAD_ROWS_DIRECT=$d"; RGL_HEAD_ROWS_DIRECT=$d python tools/trainer_time.py 2>&1 | grep '^{' | grep 'captured step, index' | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   %-60s %.4f ms per batch' % (d['workload'].split(', captured')[0], d['ms_per_batch']))"; done
} > $O/r06_head_rows.txt 2>&1
cat $O/r06_head_rows.txt
