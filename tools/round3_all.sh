#!/bin/bash
# Everything profiles/r03_* is made of, in one GPU call, at the revision given as $2 (run after the last kernel commit):
#   gpurun --timeout 2400 -- 'bash tools/round3_all.sh r03_final <git-hash>'
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r03_final}
HASH=${2:-unknown}
O=$R/gpurun_out
mkdir -p $O
cd $R
bash tools/round_profile.sh $TAG $HASH > $O/${TAG}_round_profile.log 2>&1
cp $O/${TAG}_traffic.json $R/profiles/${TAG}_traffic.json 2>/dev/null      # the bench lines below quote the record just measured
python bench.py --steps 50 --warmup 10 2>/dev/null | grep "^{" > $O/${TAG}_bench_default.json
bash tools/other_configs.sh $TAG > $O/${TAG}_other_configs.log 2>&1
bash tools/share_regime.sh $TAG > $O/${TAG}_share_regime.txt 2>&1
rm -f $O/${TAG}_timeline.md
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 256
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 512 --depth 3
bash tools/timeline.sh $O/${TAG}_timeline.md --roots 2048
python tools/train_step_time.py > $O/${TAG}_train_step.jsonl 2>/dev/null
python tools/train_step_time.py --graph 2>/dev/null | grep "^{" > $O/${TAG}_train_step_graph.jsonl
( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_tr && rocprofv3 --kernel-trace --stats -d /tmp/prof_tr -o tr -- python $R/tools/train_step_time.py > /dev/null 2>&1; \
  { echo "# $TAG: kernel trace of tools/train_step_time.py (H = 5 / 19, batch 100 / 4096; source revision $HASH)"; echo; \
    python $R/tools/rocpd_summary.py $(find /tmp/prof_tr -name "*results.db" | head -1) | head -24; } > $O/${TAG}_train_step_trace.md )
# backward pass alone: device time per call by batch size, per-scene kernel (mode 0) against the tile pipeline (mode 1)
( cd /tmp && export TMPDIR=/tmp; echo "# $TAG: device time of ONE backward call (kernel trace, value estimator and state predictor averaged; source revision $HASH)"; \
  for H in 5 19; do for B in 100 256 1024 4096; do for M in 0 1; do rm -rf /tmp/prof_bw; \
    BT_H=$H BT_B=$B BT_MODES=$M rocprofv3 --kernel-trace -d /tmp/prof_bw -o bw -- python $R/tools/backward_time.py > /dev/null 2>&1; \
    echo -n "H=$H batch=$B RGL_BACKWARD_MFMA=$M: "; python $R/tools/backward_sum.py $(find /tmp/prof_bw -name "*results.db" | head -1); done; done; done ) > $O/${TAG}_backward_by_batch.txt 2>&1
python tools/envelope_timing.py > $O/${TAG}_envelope.txt 2>&1
{ python tools/gcn_trace.py; python tools/episodes.py; } > $O/${TAG}_path_g_and_episodes.txt 2>&1
python tools/pcie_inclusive.py > $O/${TAG}_pcie_inclusive.txt 2>&1
tail -5 $O/${TAG}_other_configs.log; cat $O/${TAG}_share_regime.txt | tail -14; cat $O/${TAG}_train_step.jsonl | cut -c1-200; cat $O/${TAG}_path_g_and_episodes.txt | tail -8; head -c 600 $O/${TAG}_bench_default.json
