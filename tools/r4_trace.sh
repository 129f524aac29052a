#!/bin/bash
# kernel trace of the headline iteration -> timeline of the last iteration
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_trace
rm -rf $O; mkdir -p $O
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --nat-steps 0 > $O/bench.json 2> $O/bench.err || { echo failed; tail -5 $O/bench.err; exit 1; }
cd $R
python3 tools/iter_timeline.py $(find $O/trace -name '*kernel_trace.csv' | head -1) 150 > $O/timeline.txt
cat $O/timeline.txt
