#!/bin/bash
# dC on the row-panel kernel at full size (reads T once: -9 GB per iteration) against the tall-tile kernel, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_dcp_ab
run() { python bench.py --steps $1 --warmup 5 --nat-steps 0 --no-cpu-baseline --breakdown-steps 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2 steps $1: %.3f ms (median %.3f), frac %.3f, elbo_last %.10f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['frac'], d['elbo_last']))"; }
for i in 1 2 3 4 5; do
DGP_DCPANEL_MAX_ROWS=300000 run 50 "tall-tile dC at 10^6 rows"
DGP_DCPANEL_MAX_ROWS=4000000 run 50 "row-panel dC at 10^6 rows"
done
