#!/bin/bash
# dC on the row-panel kernel at full size (reads T once: -9 GB per iteration) against the tall-tile kernel, short and sustained runs
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_dcp_ab
run() { python bench.py --steps $1 --warmup 5 --nat-steps 0 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2 steps $1: %.2f ms (median %.2f), frac %.3f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac']))"; }
for i in 1 2 3; do
DGP_DCPANEL_MAX_ROWS=300000 run 50 "tall-tile dC at 10^6 rows"
DGP_DCPANEL_MAX_ROWS=4000000 run 50 "row-panel dC at 10^6 rows"
done
DGP_DCPANEL_MAX_ROWS=300000 run 300 "tall-tile dC at 10^6 rows"
DGP_DCPANEL_MAX_ROWS=4000000 run 300 "row-panel dC at 10^6 rows"
