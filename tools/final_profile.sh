#!/bin/bash
# Round profile of the headline workload (BASELINE configs[2]): bench line + rocprofv3 kernel trace of the same command,
# then PMC passes (separate runs, --kernel-trace only) over the value-of-children pair.
#   gpurun -- 'bash tools/final_profile.sh <tag>'   -> gpurun_out/<tag>.md  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01_x}
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 > $O/bench.log 2>&1
{
  echo "# $TAG: kernel trace + counters of the headline workload"
  echo
  echo '`rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --cpu-seconds 0` on one MI355X (tools/rocpd_summary.py over the rocpd DB; the dominant pair is launched 166 times = 60 steps x 2 levels + the bench'"'"'s stand-alone timing loop).'
  echo
  echo 'bench line of the same run:'
  echo
  echo '```'
  grep "^{" $O/bench.log
  echo '```'
  echo
  python $R/tools/rocpd_summary.py $(find $O/trace -name "*results.db" | head -1)
  echo
  echo "## PMC counters, value-of-children kernels at P = 4096 parents (one tree level of the workload)"
  echo
  echo '`rocprofv3 --kernel-trace --pmc <list> -- python tools/profile_children.py`, separate passes per counter group; SQ values are per shader engine (32 SEs); FETCH/WRITE_SIZE in KiB per dispatch.'
  echo
  echo '```'
} > $O.md
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/profile_children.py > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f children_rank1 >> $O.md; python $R/tools/pmc_summary.py $f robot_head >> $O.md; else echo "(pass $i: $grp -- no database)" >> $O.md; fi
done
echo '```' >> $O.md
rm -rf $O/trace $O/pmc*/
cat $O.md | cut -c1-200
