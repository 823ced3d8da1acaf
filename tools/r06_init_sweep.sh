#!/bin/bash
# The driver's command (`--steps 20 --warmup 5`) under different lengths of bench.py's set-up phase: does the device reach its steady
# clock before the timed region?   gpurun -- 'bash tools/r06_init_sweep.sh' -> gpurun_out/r06_init_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
out=$O/r06_init_sweep.txt; : > $out
for rep in 1 2 3; do
  for n in 40 80 160 320 640; do
    RGL_BENCH_INIT_STEPS=$n timeout 300 python bench.py --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('init %4d  steps 20: wall %.4f ms  device p10 %.4f median %.4f p90 %.4f  cold median %.4f' % ($n, d['ms_per_step'], d['step_ms_device']['p10'], d['step_ms_device']['median'], d['step_ms_device']['p90'], (d.get('step_ms_device_cold') or {}).get('median', 0)))" >> $out
  done
  RGL_BENCH_INIT_STEPS=40 timeout 300 python bench.py --steps 50 --warmup 10 --cpu-seconds 0 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('init   40  steps 50: wall %.4f ms  device p10 %.4f median %.4f p90 %.4f' % (d['ms_per_step'], d['step_ms_device']['p10'], d['step_ms_device']['median'], d['step_ms_device']['p90']))" >> $out
done
cat $out
