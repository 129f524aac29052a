#!/bin/bash
# Round-4 run 2: full GPU test suite, T store-schedule A/B (tools/tall_bench vs tall_bench_SA), rank shares.
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_run2
rm -rf $O; mkdir -p $O
cd $R
for i in 1 2; do
  timeout -k 10 60 tools/tall_bench 1000000 10 8 1 | tail -1 | sed 's/^/stores before DMA request: /' | tee -a $O/tall_ab.txt || exit 1
  timeout -k 10 60 tools/tall_bench_SA 1000000 10 8 1 | tail -1 | sed 's/^/stores behind DMA request: /' | tee -a $O/tall_ab.txt || exit 1
done
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
bash tools/shard_sizes.sh > $O/shard_sizes.txt 2>&1 || { echo "shard sizes failed"; tail $O/shard_sizes.txt; exit 1; }
cat $O/shard_sizes.txt
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
