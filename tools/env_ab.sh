#!/bin/bash
# same-box A/B of an environment switch over the sizes that matter:  bash tools/env_ab.sh <VAR=value>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/envab; O=gpurun_out/envab/ab.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | in-search children %s' % (d['ms_per_step'], d['step_ms_device']['median'], ['%.1f' % (1e3*x) for x in r['in_search_children_ms_by_level']]))"; }
for rep in 1 2; do
for sw in "" "$1"; do
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --gpus 1 --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | line "2048 roots   [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "256 roots    [$sw]" >> $O
  env $sw RGL_BENCH_NO_F32_LINE=1 python bench.py --cpu-seconds 0 --roots 512 --depth 3 --steps 100 2>/dev/null | line "512 roots D3 [$sw]" >> $O
done
done
cat $O
