#!/bin/bash
# per-step kernel timeline of a bench.py workload:  bash tools/timeline.sh <out-file> <bench flags...>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$(cd $R && realpath -m $1); shift; mkdir -p $(dirname $OUT)
rm -rf /tmp/prof_tl
RGL_BENCH_NO_F32_LINE=1 rocprofv3 --kernel-trace -d /tmp/prof_tl -o pc -- python $R/bench.py --steps 30 --warmup 5 --cpu-seconds 0 --graph off "$@" > /tmp/prof_tl.log 2>&1
{
echo "## bench.py $@"
grep "^{" /tmp/prof_tl.log | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('ms_per_step %.4f  device median %.4f  evals/s %.4g  roofline frac %.3f' % (r['ms_per_step'], r['step_ms_device']['median'], r['value'], r['roofline']['frac']))"
python $R/tools/timeline.py $(find /tmp/prof_tl -name "*results.db" | head -1)
echo
} >> $OUT
