#!/bin/bash
# L2 behaviour of the five large kernels alone (tools/tall_bench, 10^6 rows, D = 8): requests, hits, misses and the memory-side reads / writes
# behind them (TCC counters, summed over the 8 XCDs x 16 channels).  One rocprofv3 --pmc pass per counter group, program directly after --.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_l2; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/a -- $R/tools/tall_bench 1000000 2 8 61 > $O/a.log 2>&1 || { echo "pass a failed"; tail -5 $O/a.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_READ_sum TCC_WRITE_sum --kernel-trace --output-format csv -d $O/b -- $R/tools/tall_bench 1000000 2 8 61 > $O/b.log 2>&1 || { echo "pass b failed"; tail -5 $O/b.log; }
cd $R
python3 tools/pmc_table.py $O/a > $O/l2_req_hit_miss.txt 2>&1; cat $O/l2_req_hit_miss.txt | cut -c1-200
python3 tools/pmc_table.py $O/b > $O/l2_ea.txt 2>&1; cat $O/l2_ea.txt | cut -c1-220
