#!/bin/bash
# same-box check of the contraction modes: the bf16x6 tests, then f32 / bf16x6 bench lines back to back
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16x6" > $O/mode_ab_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/mode_ab_tests.log | cut -c1-300
for c in f32 bf16x6 bf16x6; do
  RGL_BENCH_NO_F32_LINE=1 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --contraction $c 2>/dev/null | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$c', 'ms_per_step %.4f device median %.4f roofline frac %.3f launch_ms %.4f levels %s' % (r['ms_per_step'], r['step_ms_device']['median'], r['roofline']['frac'], r['roofline']['launch_ms'], r['roofline']['in_search_children_ms_by_level']))
"
done 2>&1 | tee $O/mode_ab_bench.txt
