#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04b; O=gpurun_out/r04b
./tools/micro/mfma_4x4 > $O/mfma_4x4.txt 2>&1
python -m pytest tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -k "at_size or fast_batches or mfma_backward_at_size" 2>&1 | tail -60 > $O/pytest.log
for init in 3 40 150; do
  for rep in 1 2; do
    RGL_BENCH_INIT_STEPS=$init RGL_BENCH_NO_F16X3=1 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('init $init rep $rep: ms_per_step %.4f device p10 %.4f median %.4f p90 %.4f' % (d['ms_per_step'], d['step_ms_device']['p10'], d['step_ms_device']['median'], d['step_ms_device']['p90']))" >> $O/init_steps.txt
  done
done
cat $O/mfma_4x4.txt; cat $O/init_steps.txt; tail -5 $O/pytest.log
