#!/bin/bash
# Round profile of the headline workload (BASELINE configs[2]): bench line + rocprofv3 kernel trace of the same command, then
# PMC passes (separate runs, --kernel-trace only) over the value-of-children kernels at P = 4096 parents, and the HBM-traffic
# record bench.py quotes (stamped with the source revision).
#   gpurun -- 'bash tools/round_profile.sh <tag> <git-hash>'   -> gpurun_out/<tag>.md, gpurun_out/<tag>_traffic.json  (copy to profiles/)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_x}
HASH=${2:-unknown}
O=$R/gpurun_out/$TAG
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 > $O/bench.log 2>&1
{
  echo "# $TAG: kernel trace + counters of the headline workload (source revision $HASH)"
  echo
  echo '`rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --cpu-seconds 0` on one MI355X (tools/rocpd_summary.py over the rocpd DB).'
  echo
  echo 'bench line of the same run:'
  echo
  echo '```'
  grep "^{" $O/bench.log
  echo '```'
  echo
  python $R/tools/rocpd_summary.py $(find $O/trace -name "*results.db" | head -1)
  echo
  echo "## PMC counters, value-of-children kernels at P = 4096 parents (one tree level of the workload), in the mode that carries \`value\` (bf16x6)"
  echo
  echo '`rocprofv3 --kernel-trace --pmc <list> -- python tools/profile_children.py`, separate passes per counter group; SQ values are per shader engine (32 SEs; SQ_ACTIVE_* / *_BUSY / WAVE / WAIT counters in quad-cycles); FETCH/WRITE_SIZE in KiB per dispatch.'
  echo
  echo '```'
} > $O.md
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/profile_children.py > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then
    for k in children_fused pack_images robot_head children_rank1; do python $R/tools/pmc_summary.py $f $k >> $O.md; done
    if [ "$grp" = "FETCH_SIZE" ]; then cp $f $O/fetch.db; fi
    if [ "$grp" = "WRITE_SIZE" ]; then cp $f $O/write.db; fi
  else echo "(pass $i: $grp -- no database)" >> $O.md; fi
done
echo '```' >> $O.md
{
  echo
  echo "## The plain f32-MFMA form of the same launch (contraction f32) beside it: instruction counts, co-execution"
  echo
  echo '`rocprofv3 --kernel-trace --pmc <list> -- python tools/profile_children.py --contraction f32`'
  echo
  echo '```'
} >> $O.md
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o pmc -- python $R/tools/profile_children.py --contraction f32 > $O/pmc$i.log 2>&1
  f=$(find $O/pmc$i -name "*results.db" | head -1)
  if [ -n "$f" ]; then python $R/tools/pmc_summary.py $f children_fused >> $O.md; else echo "(pass $i: $grp -- no database)" >> $O.md; fi
done
{
  echo '```'
  echo
  echo "kernel durations of that mode (rocprofv3 --kernel-trace --stats -- python tools/profile_children.py --contraction f32):"
  echo
} >> $O.md
rocprofv3 --kernel-trace --stats -d $O/x3trace -o x3 -- python $R/tools/profile_children.py --contraction f32 > $O/x3.log 2>&1
python $R/tools/rocpd_summary.py $(find $O/x3trace -name "*results.db" | head -1) | head -6 >> $O.md
python - <<PY
import json, sqlite3, sys
sys.path.insert(0, "$R/tools")
sys.path.insert(0, "$R")
def totals(path, sym):
    db = sqlite3.connect(path)
    t = {r[0].split("_0000")[0]: r[0] for r in db.execute("select name from sqlite_master where type='table'")}
    q = ("select s.kernel_name, e.value, d.id from %s e join %s p on e.pmc_id = p.id join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id where p.symbol = '%s'" % (t["rocpd_pmc_event"], t["rocpd_info_pmc"],
                                                                   t["rocpd_kernel_dispatch"], t["rocpd_info_kernel_symbol"], sym))
    per = {}
    for name, val, did in db.execute(q):
        per.setdefault(name.split("(")[0], {}).setdefault(did, 0.0)
        per[name.split("(")[0]][did] += val
    return {k: sum(v.values()) / len(v) for k, v in per.items()}
try:
    f, w = totals("$O/fetch.db", "FETCH_SIZE"), totals("$O/write.db", "WRITE_SIZE")
    keep = [k for k in f if any(s in k for s in ("children_fused", "pack_images", "robot_head", "children_rank1"))]
    scenes = 4096 * 81
    total_bytes = sum(2 * f[k] + w.get(k, 0.0) for k in keep) * 1024
    rec = {"source": "profiles/${TAG}_kernel_stats.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_children.py, P = 4096 parents x 81 children, N = 20, L = 2)",
           "source_revision": "$HASH",
           "kernel_sources_sha256": __import__("bench").kernel_sources_digest(),
           "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced loads -> reads doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE as reported; both in KiB",
           "workload": {"N": 20, "L": 2, "A": 81}, "scenes": scenes,
           "kernels": {k: {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w.get(k, 0.0)} for k in keep},
           "bytes_per_scene": total_bytes / scenes}
    json.dump(rec, open("$R/gpurun_out/${TAG}_traffic.json", "w"), indent=2)
    print(json.dumps(rec)[:600])
except Exception as e:
    print("traffic record failed:", e)
PY
rm -rf $O
cat $R/gpurun_out/$TAG.md | cut -c1-220 | head -60
