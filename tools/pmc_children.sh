#!/bin/bash
# PMC passes (separate runs, --kernel-trace only) over the value-of-children kernels at P = 4096 parents.
#   gpurun -- 'bash tools/pmc_children.sh <tag> [kernel-name-substring ...]'   -> gpurun_out/<tag>_pmc.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_x}; shift
KERNELS=${@:-children_fused pack_images robot_head}
O=$R/gpurun_out/${TAG}_pmc
mkdir -p $O
: > $O.txt
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/profile_children.py > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then for k in $KERNELS; do python $R/tools/pmc_summary.py $f $k >> $O.txt; done; else echo "(pass $i: $grp -- no database)" >> $O.txt; fi
done
rm -rf $O
cat $O.txt
