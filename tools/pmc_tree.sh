#!/bin/bash
# PMC passes over whole tree searches (tools/profile_children.py --tree): counters of the kernels named on the command line
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02_t}; shift
KERNELS=${@:-scene_graph}
O=$R/gpurun_out/${TAG}_pmc
mkdir -p $O
: > $O.txt
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/profile_children.py --tree > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then for k in $KERNELS; do python $R/tools/pmc_summary.py $f $k >> $O.txt; done; else echo "(pass $i: $grp -- no database)" >> $O.txt; fi
done
rm -rf $O
cat $O.txt
