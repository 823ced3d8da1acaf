#!/bin/bash
# Quick iteration loop of round 6: targeted parity tests, the children kernel by parent count, the bench line, two PMC groups.
#   gpurun --timeout 1200 -- 'bash tools/r06_quick.sh <tag> [pytest -k expression]'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r06_q}
KEXPR=${2:-"bf16x6 or smoke or forward_kats or planning"}
O=$R/gpurun_out
mkdir -p $O
cd $R
{ python -m pytest tests -m gpu -x -q -k "$KEXPR" 2>&1 | tail -4
  python tools/kiter.py --contraction bf16x6 --quick --parents 256 512 2048 4096 2>&1 | grep -v "^$" | head -12
  python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-seconds 0 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bench ms_per_step %.4f  device median %.4f  children launch_ms %.4f  frac %.3f  f32_line %.4f' % (d['ms_per_step'], d['step_ms_device']['median'], d['roofline']['launch_ms'], d['roofline']['frac'], d.get('f32_mfma_line',{}).get('ms_per_step',0)))"
} > $O/${TAG}.txt 2>&1
if [ "$3" != "nopmc" ]; then
( cd /tmp && export TMPDIR=/tmp
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY"; do
    i=$((i+1)); rm -rf /tmp/q_pmc$i
    rocprofv3 --kernel-trace --pmc $grp -d /tmp/q_pmc$i -o pmc -- python $R/tools/profile_children.py > /tmp/q_pmc$i.log 2>&1
    f=$(find /tmp/q_pmc$i -name "*results.db" | head -1)
    [ -n "$f" ] && python $R/tools/pmc_summary.py $f children_fused >> $O/${TAG}.txt
  done )
fi
cat $O/${TAG}.txt
