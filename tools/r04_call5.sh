#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -8 > $O/pytest.log
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$1: ms_per_step %.4f device median %.4f | in-search children %s frac %.3f | step frac %.3f' % (d['ms_per_step'], d['step_ms_device']['median'], r['in_search_children_ms_by_level'], r['frac'], d['roofline_step']['frac']))"; }
for rep in 1 2; do
RGL_BENCH_NO_F16X3=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | line "next-children in tail" >> $O/bench.txt
RGL_FUSED_NO_NEXT_CHILDREN=1 RGL_BENCH_NO_F16X3=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | line "as before            " >> $O/bench.txt
done
RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "256 roots" >> $O/bench.txt
RGL_FUSED_NO_NEXT_CHILDREN=1 RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | line "256 roots, as before" >> $O/bench.txt
RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 512 --depth 3 --steps 100 2>/dev/null | line "512 roots D3" >> $O/bench.txt
RGL_FUSED_NO_NEXT_CHILDREN=1 RGL_BENCH_NO_F16X3=1 python bench.py --cpu-seconds 0 --roots 512 --depth 3 --steps 100 2>/dev/null | line "512 roots D3, as before" >> $O/bench.txt
rm -f $O/timeline.md
bash tools/timeline.sh $O/timeline.md --roots 2048
bash tools/timeline.sh $O/timeline.md --roots 256
tail -3 $O/pytest.log; cat $O/bench.txt; cat $O/timeline.md
