#!/bin/bash
# dispatch thresholds of the state-predictor path re-checked after round 6's kernel changes: bench device medians on ONE box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1.txt; : > $O
cd $R
run() { echo "== $1 | $2" >> $O; env $1 RGL_BENCH_NO_F32_LINE=1 python bench.py --steps 30 --warmup 5 --cpu-seconds 0 $2 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   ms_per_step %.4f  device median %.4f p10 %.4f' % (d['ms_per_step'], d['step_ms_device']['median'], d['step_ms_device']['p10']))" >> $O; }
for v in "X=0" "RGL_SCENE_SPLIT_BELOW=0" "RGL_SCENE_SPLIT_BELOW=2048" "RGL_SCENE_SPLIT_BELOW=4097" "RGL_SCENE_CHILDREN_BELOW=0" "RGL_SCENE_CHILDREN_BELOW=2500" "RGL_SCENE_CHILDREN_BELOW=4500" "X=0"; do run "$v" "--roots 2048"; done
for v in "X=0" "RGL_SCENE_SPLIT_BELOW=0" "RGL_SCENE_CHILDREN_BELOW=0" "RGL_SCENE_EMBED_INSIDE=0" "X=0"; do run "$v" "--roots 256"; run "$v" "--roots 512 --depth 3"; done
cat $O
