#!/bin/bash
# The per-GPU-share regime (what each of 8 GPUs runs in the fixed-total reading): bench lines with and without hipGraph replay.
#   gpurun -- 'bash tools/share_regime.sh <tag>'  -> gpurun_out/<tag>_share_regime.jsonl + a table on stdout
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_x}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/${TAG}_share_regime.jsonl
run() { python bench.py --steps 200 --warmup 20 --cpu-seconds 0 "$@" 2>/dev/null | grep "^{" >> $O/${TAG}_share_regime.jsonl; }
for g in off on; do
  run --roots 256 --graph $g
  run --roots 512 --graph $g
  run --roots 1024 --graph $g
  run --depth 3 --roots 512 --graph $g
  run --humans 5 --depth 1 --roots 512 --graph $g
done
run --roots 2048 --graph off --steps 50
run --roots 2048 --graph on --steps 50
run --depth 3 --roots 4096 --steps 30 --graph off
python - <<PY
import json
for l in open("$O/${TAG}_share_regime.jsonl"):
    d = json.loads(l)
    print("%-95s graph=%-5s wall %.4f ms  device median %.4f ms  %.3g evals/s  frac %.3f" % (
        d["config"]["workload"][:95], d["config"]["graph_replay"], d["ms_per_step"], d["step_ms_device"]["median"], d["value"],
        d["roofline"]["frac"]))
PY
