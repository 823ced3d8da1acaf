#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04c; O=gpurun_out/r04c
python -m pytest tests -q -m gpu -p no:cacheprovider -x 2>&1 | tail -15 > $O/pytest.log
for rep in 1 2; do
python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ms_per_step %.4f device median %.4f cold %s | in-search children %s frac %.3f standalone %.4f | step frac %.3f | f16x3 %.4f' % (d['ms_per_step'], d['step_ms_device']['median'], d['step_ms_device_cold'], r['in_search_children_ms_by_level'], r['frac'], r['standalone_launch_ms'], d['roofline_step']['frac'], d['f16x3']['ms_per_step']))" >> $O/bench.txt
done
python bench.py --cpu-seconds 0 --roots 256 --steps 200 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('256 roots: ms_per_step %.4f device median %.4f' % (d['ms_per_step'], d['step_ms_device']['median']))" >> $O/bench.txt
python bench.py --cpu-seconds 0 --roots 512 --depth 3 --steps 100 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('512 roots D3: ms_per_step %.4f device median %.4f' % (d['ms_per_step'], d['step_ms_device']['median']))" >> $O/bench.txt
tail -4 $O/pytest.log; cat $O/bench.txt
