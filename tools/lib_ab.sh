#!/bin/bash
# same-box A/B of two builds of the library: the tree's against a second one kept under ab/ (git-ignored, travels with gpurun), e.g.
#   git stash; make -C relationalgraphlearning_amd/csrc; cp relationalgraphlearning_amd/lib/librgl_hip.so ab/librgl_prev.so; git stash pop; make ...
#   bash tools/lib_ab.sh [ab/librgl_prev.so]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/libab; O=gpurun_out/libab/ab.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | in-search children %s' % (d['ms_per_step'], d['step_ms_device']['median'], ['%.1f' % (1e3*x) for x in r['in_search_children_ms_by_level']]))"; }
for rep in 1 2; do
for sw in "X=tree" "RGL_HIP_LIBRARY=$PWD/${1:-ab/librgl_prev.so}"; do
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --gpus 1 --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | line "c2 2048 roots   [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "c2 256 roots    [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 --contraction f16 2>/dev/null | line "c4 f16 256 roots [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 2>/dev/null | line "c4 f32 256 roots [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --humans 39 --layers 3 --roots 256 --steps 100 2>/dev/null | line "N=40 L=3 f32 256 [$sw]" >> $O
done
done
cat $O
