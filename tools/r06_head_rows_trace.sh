#!/bin/bash
# rocprofv3 kernel-trace durations of the value head's row kernels inside the trainer's captured batch-100 step, for each library
# given (default: the product's with RGL_HEAD_ROWS_DIRECT=1 and =0).  usage: tools/r06_head_rows_trace.sh [lib.so ...]
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/tn && mkdir -p /tmp/tn
  rocprofv3 --kernel-trace --stats -d /tmp/tn -o tn -- python $R/tools/trainer_trace.py 5 10 > /tmp/tn/run.log 2>&1
  python - "$1" <<'PY'
import glob, sqlite3, sys
db = sqlite3.connect(glob.glob('/tmp/tn/**/*_results.db', recursive=True)[0])
c = db.cursor()
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by count(*) desc").fetchall()
for name, cnt, avg, mn in rows:
    if 'mlp_rows_kernel' in name or 'head_rows' in name or 'robot_head' in name:
        print("%-34s %6d calls  avg %.2f us  min %.2f us  %s" % (sys.argv[1], cnt, avg / 1e3, mn / 1e3, name[20:60]))
PY
}
if [ $# -eq 0 ]; then
  RGL_HEAD_ROWS_DIRECT=1 run "direct (head_rows_kernel)"
  RGL_HEAD_ROWS_DIRECT=0 run "staged (mlp_rows_kernel)"
else
  for lib in "$@"; do RGL_HIP_LIBRARY=$R/$lib run "$(basename $lib)"; done
fi
