#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_pf
for i in 1 2; do
timeout -k 10 60 tools/tall_bench 1000000 10 8 64 | tail -1 | sed 's/^/fragments 2 units ahead: /' | tee -a gpurun_out/r4_pf/t.txt
timeout -k 10 60 tools/tall_bench_PF4 1000000 10 8 64 | tail -1 | sed 's/^/fragments 4 units ahead: /' | tee -a gpurun_out/r4_pf/t.txt
done
