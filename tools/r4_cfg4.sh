#!/bin/bash
# where config 4's shard iteration goes (kernel stats), and what the Mp = 256 Gram kernel would do on the same
# points if the 512 x 512 triangle were cut into four 256 x 256 triangles (two diagonal: weighted one-source form; the rectangle:
# two lower triangles of the two-source form) - a timing proxy on the existing kernels, no new code.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=$R/gpurun_out/r4_cfg4; mkdir -p $O; export TMPDIR=/tmp
cat $O/gram_proxy.txt 2>/dev/null;
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o cfg4 -- python3 tools/cfg4_profile.py > $O/run.log 2>&1
tail -3 $O/run.log
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "stats: $f"; [ -n "$f" ] && cp $f $O/cfg4_kernel_stats.csv && head -25 $f | cut -c1-200
rm -rf $O/prof
