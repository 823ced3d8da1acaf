#!/bin/bash
# Path G (GCN.predict_batch, 2048 roots x 81 rotated scenes): kernel trace + PMC passes (separate runs, --kernel-trace only) of the
# gcn_* / graph-forward kernels it launches.   gpurun -- 'bash tools/path_g_profile.sh <tag>'   -> gpurun_out/<tag>_path_g_profile.md
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05}
O=$R/gpurun_out/${TAG}_path_g_profile
mkdir -p $O
OUT=$O.md
{
echo "# $TAG: path G, GCN.predict_batch at 2048 roots (tools/gcn_trace.py), kernel trace + counters"
echo
rm -rf /tmp/prof_pg
rocprofv3 --kernel-trace --stats -d /tmp/prof_pg -o pg -- python $R/tools/gcn_trace.py > /tmp/pg.log 2>&1
echo '```'; grep -v "^{" /tmp/pg.log | grep "path G"; echo '```'; echo
python $R/tools/rocpd_summary.py $(find /tmp/prof_pg -name "*results.db" | head -1) | head -30
echo
echo "## counters (per launch averages over both crowd sizes; separate passes)"
echo
echo '```'
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  RGL_GCN_TRACE_ONLY_H=19 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/gcn_trace.py > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f; else echo "(pass $i: $grp -- no database)"; fi
done
echo '```'
} > $OUT 2>&1
rm -rf $O
head -40 $OUT
