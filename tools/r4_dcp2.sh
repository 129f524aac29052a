#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out/r4_dcp2
for n in 250000 500000; do timeout -k 10 60 tools/tall_bench $n 10 8 72 | tail -2 | sed "s/^/D=8 $n rows  /" | tee -a gpurun_out/r4_dcp2/t.txt; done
for d in 8 1 3; do timeout -k 10 60 tools/tall_bench 1000000 10 $d 72 | tail -2 | sed "s/^/D=$d  /" | tee -a gpurun_out/r4_dcp2/t.txt; done
timeout -k 10 60 tools/tall_bench 124928 20 8 72 | tail -2 | sed "s/^/D=8 124928 rows  /" | tee -a gpurun_out/r4_dcp2/t.txt
timeout -k 10 60 tools/tall_bench 12544 50 8 72 | tail -2 | sed "s/^/D=8 12544 rows  /" | tee -a gpurun_out/r4_dcp2/t.txt
