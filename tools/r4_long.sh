#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_long
timeout -k 10 300 python tools/long_run.py 2>&1 | tee gpurun_out/r4_long/long_run.txt
timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --nat-steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps', d['steps'], 'mean', round(d['ms_per_step'],3), 'ms median', round(d['ms_per_step_median'],3), 'ms min/max', [round(x,3) for x in d['ms_per_step_min_max']], 'ms ELBO after 310 iterations', d['elbo_last'], '| then 50 nat-adam iterations:', round(d['nat_adam_ms_per_iteration'],2), 'ms each, ELBO', d['nat_adam_elbo_last'])" | tee -a gpurun_out/r4_long/long_run.txt
