#!/bin/bash
# bench lines of the other BASELINE configurations and sizes (one JSON line each) + pair-by-size tables + single-decision latency
#   gpurun -- 'bash tools/other_configs.sh <tag>'  -> gpurun_out/<tag>_other_configs.jsonl, <tag>_pair_by_size_*.txt, <tag>_latency.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_x}
O=$R/gpurun_out
mkdir -p $O
cd $R
: > $O/${TAG}_other_configs.jsonl
run() { python bench.py --steps 100 --warmup 10 --cpu-seconds 0 "$@" 2>/dev/null | grep "^{" >> $O/${TAG}_other_configs.jsonl; }
run --humans 4 --depth 1 --roots 512
run --humans 5 --depth 1 --roots 512
run --depth 3 --roots 512
run --roots 16384 --steps 30
run --scaling strong --total-roots 4096 --depth 3 --steps 30
run --humans 49 --layers 3 --roots 256
run --humans 49 --layers 3 --roots 256 --contraction f16
run --contraction f32 --steps 50
run --depth 3 --roots 512 --contraction f32

python tools/kiter.py > $O/${TAG}_pair_by_size_dispatch.txt 2>&1
RGL_CHILDREN_FUSED=1 python tools/kiter.py --quick > $O/${TAG}_pair_by_size_fused.txt 2>&1
RGL_CHILDREN_TWO_STAGE=1 python tools/kiter.py --quick > $O/${TAG}_pair_by_size_two_stage.txt 2>&1
python tools/latency.py > $O/${TAG}_latency.txt 2>&1
python - <<PY
import json
for l in open("$O/${TAG}_other_configs.jsonl"):
    d = json.loads(l)
    print("%-100s %.4f ms  %.3g evals/s  frac %.3f" % (d["config"]["workload"][:100], d["ms_per_step"], d["value"], d["roofline"]["frac"]))
PY
grep "pair P\|search" $O/${TAG}_pair_by_size_*.txt; cat $O/${TAG}_latency.txt | grep predict
