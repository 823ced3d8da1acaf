#!/bin/bash
# PMC passes over the value-of-children pair on the BASELINE configs[4] shape (N = 50, L = 3), fp32 and f16 contractions.
#   gpurun -- 'bash tools/deep_pmc.sh [parents]'   -> gpurun_out/deep_pmc.md
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
P=${1:-512}
O=$R/gpurun_out/deep_pmc
mkdir -p $O
{
  echo "## PMC counters, children_deep_kernel + robot_head_kernel at P = $P parents, N = 50, L = 3"
  echo
  echo '`rocprofv3 --kernel-trace --pmc <list> -- python tools/profile_children.py --parents '$P' --humans 49 --layers 3 [--contraction f16]`, separate passes per counter group; SQ values per shader engine (32 SEs); FETCH/WRITE_SIZE in KiB per dispatch.'
} > $O.md
for v in f32 f16; do
  echo >> $O.md; echo "### contraction $v" >> $O.md; echo >> $O.md; echo '```' >> $O.md
  i=0
  for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $grp -d $O/$v$i -o pmc -- python $R/tools/profile_children.py --parents $P --humans 49 --layers 3 --contraction $v > $O/$v$i.log 2>&1
    f=$(find $O/$v$i -name "*results.db" | head -1)
    if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f children_deep >> $O.md; else echo "(pass $i: $grp -- no database)" >> $O.md; fi
  done
  echo '```' >> $O.md
done
rm -rf $O/f32*/ $O/f16*/
cat $O.md | cut -c1-160
