#!/bin/bash
# kernel trace of whole tree searches (bench.py workload): per-kernel durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_ks
rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o pc -- python $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 "$@" > /tmp/prof_ks.log 2>&1
grep "^{" /tmp/prof_ks.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms_per_step %.4f  evals/s %.4g  roofline frac %.3f  pair launch ms %.4f' % (r['ms_per_step'], r['value'], r['roofline']['frac'], r['roofline']['launch_ms']))"
python $R/tools/rocpd_summary.py $(find /tmp/prof_ks -name "*results.db" | head -1) | head -14 | cut -c1-170
