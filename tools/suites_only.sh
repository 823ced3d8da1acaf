#!/bin/bash
# The three suite runs and smoke() of tools/final_checks.sh on their own (after test-only commits: the kernel records stay valid while
# bench.py's kernel_sources_digest matches profiles/<tag>_traffic.json):  gpurun -- 'bash tools/suites_only.sh <tag> <git-hash>'
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
  RGL_CONTRACT_F32_AS=bf16x6 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -170; } > $O/${TAG}_suite_under_bf16x6.txt
tail -2 $O/${TAG}_suite_under_bf16x6.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
{ echo "# RGL_DEBUG_POISON_WORKSPACES=1 python -m pytest tests -m gpu -q   (workspaces and output slabs pre-filled with NaN patterns), source revision $HASH"; RGL_DEBUG_POISON_WORKSPACES=1 timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -3; } > $O/${TAG}_suite_poisoned_workspaces.txt
tail -1 $O/${TAG}_suite_poisoned_workspaces.txt
python -c "
import sys; sys.path.insert(0, '.')
import bench, json
t = json.load(open('profiles/${TAG}_traffic.json'))
print('kernel sources digest', bench.kernel_sources_digest()[:16], 'traffic record', str(t.get('kernel_sources_digest', t.get('kernel_sources_sha256', '')))[:16])"
