#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "bf16x6" > $O/r05_e_tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/r05_e_tests.log
RGL_CONTRACT_F32_AS=bf16x6 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "planning_kats or tree_vs_batched or at_size or properties or state_predictor" > $O/r05_e_tests2.log 2>&1
echo "tests under mode rc=$?"; tail -4 $O/r05_e_tests2.log
for c in f32 bf16x6 f32 bf16x6; do
  RGL_BENCH_NO_F32_LINE=1 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 --contraction $c 2>/dev/null | grep "^{" | python -c "
import sys, json
r = json.loads(sys.stdin.read())
print('$c', 'ms_per_step %.4f device median %.4f roofline frac %.3f peak %.1f launch_ms %.4f' % (r['ms_per_step'], r['step_ms_device']['median'], r['roofline']['frac'], r['roofline']['peak'], r['roofline']['launch_ms']))
"
done 2>&1 | tee $O/r05_e_bench_ab.txt
rm -f $O/r05_e_timeline.md
bash tools/timeline.sh $O/r05_e_timeline.md --roots 2048
bash tools/timeline.sh $O/r05_e_timeline.md --roots 256
grep -v "^$" $O/r05_e_timeline.md | head -30
