#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04ab; O=gpurun_out/r04ab/deep.txt; : > $O
python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "H49 or 49 or deep or f16_contraction or configs4 or configs or other_similarities or tile_kernel_variant or forward_kats" 2>&1 | tail -4 >> $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | in-search children %s frac %.3f' % (d['ms_per_step'], d['step_ms_device']['median'], ['%.1f' % (1e3*x) for x in r['in_search_children_ms_by_level']], r['frac']))"; }
for rep in 1 2; do
for sw in "X=1" "RGL_DEEP_T4=0"; do
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 2>/dev/null | line "c4 f32 256 roots [$sw]" >> $O
  env $sw RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 2048 --steps 30 2>/dev/null | line "c4 f32 2048 roots [$sw]" >> $O
done
env RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --steps 100 --contraction f16 2>/dev/null | line "c4 f16 256 roots" >> $O
env RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --humans 39 --layers 3 --roots 256 --steps 100 2>/dev/null | line "N=40 L=3 f32 256 roots (non-T4 form)" >> $O
done
cat $O
