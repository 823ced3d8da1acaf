#!/bin/bash
# round 5, first GPU call: the whole GPU suite on the new root-clip convention, the bf16-split micro-benchmark, the bench line,
# a timeline of the headline step (baseline for the state-predictor work) and the path-G profile.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r05_a_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -5 $O/r05_a_gpu_tests.log
timeout 300 ./tools/micro/bf16x3_split > $O/r05_micro_bf16x3_split.txt 2>&1
cat $O/r05_micro_bf16x3_split.txt
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/r05_a_bench_driver_style.json
head -c 900 $O/r05_a_bench_driver_style.json; echo
rm -f $O/r05_a_timeline.md
bash tools/timeline.sh $O/r05_a_timeline.md --roots 2048
bash tools/timeline.sh $O/r05_a_timeline.md --roots 256
cat $O/r05_a_timeline.md | head -60
bash tools/path_g_profile.sh r05_a
