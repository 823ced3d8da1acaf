#!/bin/bash
# same-box sweep of dispatch switches (each line: one bench.py run)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04ab; O=gpurun_out/r04ab/sweep.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | children %s' % (d['ms_per_step'], d['step_ms_device']['median'], ['%.1f' % (1e3*x) for x in r['in_search_children_ms_by_level']]))"; }
run() { env $1 RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 $2 2>/dev/null | line "$2 [$1]" >> $O; }
for sw in "X=1" "RGL_FUSED_INLINE_PARTIAL=0" "RGL_FUSED_INLINE_PARTIAL=1" "RGL_SCENE_SPLIT_BELOW=0" "RGL_SCENE_SPLIT_BELOW=8192" "RGL_FUSED_G=1" "RGL_FUSED_G=2" "RGL_FUSED_G=3" "X=2"; do
  run "$sw" "--gpus 1 --steps 50 --warmup 10"
done
for sw in "X=1" "RGL_FUSED_INLINE_PARTIAL=0" "RGL_FUSED_INLINE_PARTIAL=1" "RGL_SCENE_SPLIT_BELOW=0" "RGL_SCENE_EMBED_INSIDE=0" "RGL_FUSED_G=1" "RGL_FUSED_G=2" "RGL_FUSED_G=3" "RGL_CHILDREN_TWO_STAGE=1" "X=2"; do
  run "$sw" "--roots 256 --steps 200"
done
for sw in "X=1" "RGL_FUSED_INLINE_PARTIAL=0" "RGL_FUSED_INLINE_PARTIAL=1" "RGL_SCENE_SPLIT_BELOW=0" "RGL_SCENE_EMBED_INSIDE=0" "RGL_FUSED_G=1" "RGL_FUSED_G=2" "RGL_FUSED_G=3" "X=2"; do
  run "$sw" "--roots 512 --depth 3 --steps 100"
done
cat $O
