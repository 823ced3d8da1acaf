#!/bin/bash
# bench.py's multi-rank step path with ONE rank over RCCL (torch.distributed.run --nproc-per-node 1) beside the plain single-process
# line: what the exchange path adds to a step.   gpurun -- 'bash tools/r06_rccl_one_rank.sh' -> gpurun_out/r06_rccl_one_rank.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out; mkdir -p $O
cd $R
p() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); m=d.get('multi_gpu') or {}
print('%-34s ms_per_step %.4f  device median %.4f  host: search enqueue %s ms, exchange %s ms per step' % ('$1', d['ms_per_step'], d['step_ms_device']['median'], (m.get('search_ms_per_step_by_rank') or ['-'])[0], (m.get('exchange_ms_per_step_by_rank') or ['-'])[0]))"; }
T="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511"
{
for roots in 256 512 2048; do
  timeout 300 python bench.py --gpus 1 --steps 50 --warmup 10 --roots $roots --cpu-seconds 0 2>/dev/null | grep '^{' | p "plain, roots $roots"
  timeout 300 $T bench.py --gpus 1 --steps 50 --warmup 10 --roots $roots --cpu-seconds 0 2>/dev/null | grep '^{' | p "1-rank RCCL, roots $roots"
  timeout 300 $T bench.py --gpus 1 --steps 50 --warmup 10 --roots $roots --cpu-seconds 0 --graph off 2>/dev/null | grep '^{' | p "1-rank RCCL, --graph off, roots $roots"
done
} > $O/r06_rccl_one_rank.txt 2>&1
cat $O/r06_rccl_one_rank.txt
