#!/bin/bash
# round 5, second GPU call: the bf16x6 mode -- its tests, then f32 vs bf16x6 bench lines back to back on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16x6 or root_clip or growing" > $O/r05_b_tests.log 2>&1
echo "tests rc=$?"; tail -15 $O/r05_b_tests.log
for c in f32 bf16x6 f32 bf16x6; do
  RGL_BENCH_NO_F16X3=1 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --contraction $c 2>/dev/null | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$c', 'ms_per_step %.4f device median %.4f roofline frac %.3f peak %.1f launch_ms %.4f' % (r['ms_per_step'], r['step_ms_device']['median'], r['roofline']['frac'], r['roofline']['peak'], r['roofline']['launch_ms']))
"
done 2>&1 | tee $O/r05_b_bench_ab.txt
