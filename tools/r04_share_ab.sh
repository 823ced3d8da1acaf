#!/bin/bash
# same-box A/B on the small-share workloads (where the state predictor's scene kernel embeds its own rows):
#   bash tools/r04_share_ab.sh <VAR=value>        e.g. RGL_HIP_LIBRARY=$PWD/ab/librgl_prev.so
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04share; O=gpurun_out/r04share/ab.txt; : > $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: ms_per_step %.4f device median %.4f' % (d['ms_per_step'], d['step_ms_device']['median']))"; }
for rep in 1 2; do
for sw in "X=default" "$1"; do
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "c2 256 roots       [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 512 --depth 3 --steps 100 2>/dev/null | line "c3 share 512 D3    [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 5 --depth 1 --roots 512 --steps 300 2>/dev/null | line "c1 N=6 512 roots   [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 --contraction f16 2>/dev/null | line "c4 f16 256 roots   [$sw]" >> $O
done
done
cat $O
