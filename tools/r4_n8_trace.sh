#!/bin/bash
# kernel trace of a rank's share of 8 GPUs (N = 12496) in the three-stage form, full timeline (every launch >= 3 us)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_n8
rm -rf $O; mkdir -p $O
cd /tmp
DGP_BENCH_PATH=partial timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --N 12496 --steps 6 --warmup 2 --no-cpu-baseline --nat-steps 0 --breakdown-steps 6 > $O/bench.json 2> $O/bench.err || { echo failed; tail -5 $O/bench.err; exit 1; }
cd $R
python3 tools/iter_timeline.py $(find $O/trace -name "*kernel_trace.csv" | head -1) 3 8 > $O/n8_timeline.txt
tail -3 $O/n8_timeline.txt
