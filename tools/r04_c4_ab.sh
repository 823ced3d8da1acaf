#!/bin/bash
# same-box A/B of an environment switch on BASELINE configs[4] (per-GPU share, f32 and f16 contraction; 2048 roots f32) and a 3-layer
# N = 20 case:  bash tools/r04_c4_ab.sh <VAR=value>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04c4; O=gpurun_out/r04c4/ab.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | in-search children %s frac %.3f' % (d['ms_per_step'], d['step_ms_device']['median'], ['%.1f' % (1e3*x) for x in r['in_search_children_ms_by_level']], r['frac']))"; }
for rep in 1 2; do
for sw in "X=default" "$1"; do
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 2>/dev/null | line "c4 f32 256 roots  [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 --contraction f16 2>/dev/null | line "c4 f16 256 roots  [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 2048 --steps 30 2>/dev/null | line "c4 f32 2048 roots [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 19 --layers 3 --roots 256 --steps 100 2>/dev/null | line "N=20 L=3 f32 256  [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "c2 256 roots      [$sw]" >> $O
done
done
cat $O
