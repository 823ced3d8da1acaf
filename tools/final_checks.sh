#!/bin/bash
# What the round's records are made of, in ONE GPU call (after the last kernel commit):
#   gpurun --timeout 3300 -- 'bash tools/final_checks.sh <tag> <git-hash>'
# the GPU suite as it is, the same suite with every f32 search forced into the bf16x6 mode (admission run), smoke(), round_all.sh
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_final}
HASH=${2:-unknown}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -q > $O/${TAG}_gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -2 $O/${TAG}_gpu_tests.log
{ echo "# RGL_CONTRACT_F32_AS=bf16x6 python -m pytest tests -m gpu -q   (every search that asks for f32 runs RGL_CONTRACT_BF16X6;"
  echo "# the f32 bounds of the suite -- north star 1e-4, regression level REG_F32 = 1e-6 -- are held against it), source revision $HASH"
  RGL_CONTRACT_F32_AS=bf16x6 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -150; } > $O/${TAG}_suite_under_bf16x6.txt
tail -2 $O/${TAG}_suite_under_bf16x6.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
{ echo "# RGL_DEBUG_POISON_WORKSPACES=1 python -m pytest tests -m gpu -q   (workspaces and output slabs pre-filled with NaN patterns), source revision $HASH"; RGL_DEBUG_POISON_WORKSPACES=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3; } > $O/${TAG}_suite_poisoned_workspaces.txt
tail -1 $O/${TAG}_suite_poisoned_workspaces.txt
bash tools/round_all.sh $TAG $HASH
