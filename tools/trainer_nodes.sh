#!/bin/bash
# The public trainer's captured batch-100 step: wall time per batch (tools/trainer_time.py) and the kernels of one replayed step
# (rocprofv3 kernel trace of tools/trainer_trace.py; calls / batches run = nodes per step).  usage: tools/trainer_nodes.sh <tag>
TAG=${1:-trainer}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$REPO/gpurun_out; mkdir -p $O
cd $REPO
python tools/trainer_time.py 2>&1 | grep '^{' > $O/${TAG}_trainer_api.jsonl
RGL_TRAINER_FUSED_ADAM=0 python tools/trainer_time.py 2>&1 | grep '^{' | grep "captured step, index" > $O/${TAG}_trainer_api_foreach_adam.jsonl
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tn && mkdir -p /tmp/tn
rocprofv3 --kernel-trace --stats -d /tmp/tn -o tn -- python $REPO/tools/trainer_trace.py 5 10 > /tmp/tn/run.log 2>&1
grep '^{' /tmp/tn/run.log > $O/${TAG}_trainer_step_nodes.run.json
python - "$O/${TAG}_trainer_step_nodes.md" <<'PY'
import glob, json, sqlite3, sys
db = sqlite3.connect(glob.glob('/tmp/tn/**/*_results.db', recursive=True)[0])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
run = json.loads(open('/tmp/tn/run.log').read().split('\n{')[-1].join(['{', '']) if False else [l for l in open('/tmp/tn/run.log') if l.startswith('{')][-1])
n = run["batches_run"]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by count(*) desc").fetchall()
with open(sys.argv[1], 'w') as f:
    f.write("kernels of MPRLTrainer.optimize_batch's captured step (H=5, batch 100), %d batches run: calls per batch, average duration "
            "under the tracer (launch-sized kernels read ~4.4 us there; ~2.5-3 us in an untraced replay)\n\n" % n)
    f.write("| calls / batch | avg us | kernel |\n|---|---|---|\n")
    tot = 0.0
    for name, cnt, avg in rows:
        if cnt >= 0.9 * n and 'copyBuffer' not in name:
            f.write("| %.2f | %.2f | `%s` |\n" % (cnt / n, avg / 1e3, name[:150]))
            tot += cnt / n
    f.write("\nnodes per step: %.1f (+1 index copy; the tracer's other copyBuffer calls are the set-up's host-to-device pushes)\n" % tot)
PY
tail -2 $O/${TAG}_trainer_step_nodes.md
