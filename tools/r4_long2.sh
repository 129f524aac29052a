#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_long
timeout -k 10 300 python bench.py --steps 300 --warmup 10 --no-cpu-baseline --nat-steps 50 > gpurun_out/r4_long/bench300.json 2> gpurun_out/r4_long/bench300.err; echo "rc=$?"
tail -5 gpurun_out/r4_long/bench300.err
head -c 600 gpurun_out/r4_long/bench300.json
