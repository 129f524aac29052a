#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r4_ab_ss
for i in 1 2 3; do
  timeout -k 10 60 tools/tall_bench 1000000 10 8 1 | tail -1 | sed 's/^/16-byte halves of a sector per store:  /' | tee -a gpurun_out/r4_ab_ss/ab.txt || exit 1
  timeout -k 10 60 tools/tall_bench_SS 1000000 10 8 1 | tail -1 | sed 's/^/whole sectors per store instruction:  /' | tee -a gpurun_out/r4_ab_ss/ab.txt || exit 1
done
