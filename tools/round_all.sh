#!/bin/bash
# Everything profiles/<tag>_* is made of, in one GPU call, at the revision given as $2 (run after the last kernel commit):
#   gpurun --timeout 2700 -- 'bash tools/round_all.sh r05_final <git-hash>'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r05_final}
HASH=${2:-unknown}
O=$R/gpurun_out
mkdir -p $O
cd $R
# 1. kernel trace + PMC passes + the HBM-traffic record; the bench line embedded in the summary is taken AFTER the record exists
bash tools/round_profile.sh $TAG $HASH > $O/${TAG}_round_profile.log 2>&1
cp $O/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json 2>/dev/null
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | grep "^{" > $O/${TAG}_bench_driver_style.json      # what the driver runs
python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/${TAG}_bench_default.json
python - <<PY
import json, re
p = "$O/$TAG.md"
s = open(p).read()
line = open("$O/${TAG}_bench_default.json").read().strip()
s = re.sub(r"bench line of the same run:\n\n\`\`\`\n.*?\n\`\`\`", "bench line (python bench.py --steps 50 --warmup 10, after the traffic record below was written):\n\n\`\`\`\n" + line.replace("\\\\", "\\\\\\\\") + "\n\`\`\`", s, count=1, flags=re.S)
open(p, "w").write(s)
PY
# 2. the other BASELINE configurations, the share regime, timelines
bash tools/other_configs.sh $TAG > $O/${TAG}_other_configs.log 2>&1
bash tools/share_regime.sh $TAG > $O/${TAG}_share_regime.txt 2>&1
rm -f $O/${TAG}_timeline.md
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 256
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 512 --depth 3
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 2048
bash tools/timeline.sh $O/${TAG}_timeline.md --humans 49 --layers 3 --roots 256
bash tools/timeline.sh $O/${TAG}_timeline.md --humans 49 --layers 3 --roots 256 --contraction f16
# 3. configs[4]: kernel trace of the bench commands + PMC passes of the deep kernel / head kernel, f32 and f16 (VERDICT r3 missing 2)
( cd /tmp && export TMPDIR=/tmp
  { echo "# $TAG: BASELINE configs[4] per-GPU share (N = 50, L = 3, D = 2, w = 2, 256 roots), source revision $HASH"; echo;
    for c in f32 f16; do rm -rf /tmp/prof_c5; RGL_BENCH_NO_F32_LINE=1 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -o c5 -- python $R/bench.py --steps 50 --warmup 10 --cpu-seconds 0 --humans 49 --layers 3 --roots 256 --contraction $c > /tmp/c5.log 2>&1
      echo "== contraction $c"; echo; echo '```'; grep "^{" /tmp/c5.log; echo '```'; echo; python $R/tools/rocpd_summary.py $(find /tmp/prof_c5 -name "*results.db" | head -1) | head -12; echo; done; } > $O/${TAG}_c5_kernel_stats.md )
bash tools/deep_pmc.sh 512 > /dev/null 2>&1
cat $O/deep_pmc.md >> $O/${TAG}_c5_kernel_stats.md
# 4. training: the public trainer, the raw step, the backward by batch
bash tools/trainer_nodes.sh $TAG > /dev/null 2>&1          # ${TAG}_trainer_api.jsonl (+ _foreach_adam), ${TAG}_trainer_step_nodes.md
python tools/micro/captured_step_repeatability.py 10 > $O/${TAG}_captured_step_repeatability.txt 2>&1
python tools/train_step_time.py 2>/dev/null | grep "^{" > $O/${TAG}_train_step.jsonl
python tools/train_step_time.py --graph 2>/dev/null | grep "^{" > $O/${TAG}_train_step_graph.jsonl
# 5. path G (with its roofline line), closed-loop episodes, PCIe-inclusive step, single-decision latency rides in other_configs.sh
{ python tools/gcn_trace.py; python tools/episodes.py; } > $O/${TAG}_path_g_and_episodes.txt 2>&1
python tools/pcie_inclusive.py > $O/${TAG}_pcie_inclusive.txt 2>&1
bash tools/path_g_profile.sh $TAG > /dev/null 2>&1                 # kernel trace + counter passes of the path-G launches
python tools/micro/graph_gap.py > $O/${TAG}_micro_graph_gap.txt 2>&1
cat $O/${TAG}_micro_graph_gap.txt
tail -12 $O/${TAG}_other_configs.log; tail -14 $O/${TAG}_share_regime.txt; cat $O/${TAG}_trainer_api.jsonl | cut -c1-220; grep "^{" $O/${TAG}_path_g_and_episodes.txt | cut -c1-300; head -c 700 $O/${TAG}_bench_driver_style.json
