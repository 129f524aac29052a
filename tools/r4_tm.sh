#!/bin/bash
# Clock records of the forward T kernel (profiles/r4_tall_positions.txt): build tools/tall_bench with -DTL_TIMING (and, for the spread
# form of the stores, -DTL_STORE_SPREAD=1) HERE (hipcc cross-compiles), then on the GPU box:  bash tools/r4_tm.sh run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
OBJ="dgp-toolbox_amd/csrc/build/gemm_f64.o dgp-toolbox_amd/csrc/build/gemm_wide.o dgp-toolbox_amd/csrc/build/gemm_gram.o dgp-toolbox_amd/csrc/build/gemm_tall.o dgp-toolbox_amd/csrc/build/gemm_tallu.o dgp-toolbox_amd/csrc/build/gemm_small.o dgp-toolbox_amd/csrc/build/gemm_mid.o dgp-toolbox_amd/csrc/build/chain.o dgp-toolbox_amd/csrc/build/gemm_gpanel.o dgp-toolbox_amd/csrc/build/gemm_dcpanel.o dgp-toolbox_amd/csrc/build/points.o"
if [ "$1" != run ]; then
  for v in "TM -DTL_TIMING" "TMS -DTL_TIMING -DTL_STORE_SPREAD=1"; do
    set -- $v; n=$1; shift
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC "$@" -I dgp-toolbox_amd/csrc -I include -c tools/tall_bench.hip -o /tmp/tall_bench_$n.o &&
    hipcc --offload-arch=gfx950 -o tools/tall_bench_$n /tmp/tall_bench_$n.o $OBJ || exit 1
  done
  exit 0
fi
mkdir -p gpurun_out/r4_tm
timeout -k 10 120 tools/tall_bench_TM 1000000 5 8 261 > gpurun_out/r4_tm/d8.txt 2>&1 &&
timeout -k 10 120 tools/tall_bench_TMS 1000000 5 8 257 > gpurun_out/r4_tm/d8_spread.txt 2>&1
grep "T  =\|pair:" gpurun_out/r4_tm/d8.txt gpurun_out/r4_tm/d8_spread.txt
