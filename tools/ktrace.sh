#!/bin/bash
# kernel trace of tools/profile_children.py (value-of-children kernels at P parents): per-kernel durations
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_kt
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o pc -- python $R/tools/profile_children.py "$@" > /dev/null 2>&1
python $R/tools/rocpd_summary.py $(find /tmp/prof_kt -name "*results.db" | head -1) | head -7 | cut -c1-160
