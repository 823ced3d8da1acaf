#!/bin/bash
# same-box sweep of the embedding-row launch's grid switches (RGL_ROW_GRID_B / RGL_ROW_GRID_C / RGL_ROW_CHILDREN_FIRST), configs[2]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04row; O=gpurun_out/r04row/sweep.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f p10 %.4f' % (d['ms_per_step'], d['step_ms_device']['median'], d['step_ms_device']['p10']))"; }
for rep in 1 2; do
for sw in "X=0" "RGL_ROW_CHILDREN_FIRST=1" "RGL_ROW_GRID_B=512" "RGL_ROW_GRID_B=256" "RGL_ROW_GRID_C=1024" "RGL_ROW_GRID_C=512" "RGL_ROW_GRID_C=256" "RGL_ROW_CHILDREN_FIRST=1 RGL_ROW_GRID_B=512" "RGL_ROW_CHILDREN_FIRST=1 RGL_ROW_GRID_C=512"; do
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --gpus 1 --steps 100 --warmup 10 --cpu-seconds 0 2>/dev/null | line "2048 roots [$sw]" >> $O
done
done
cat $O
