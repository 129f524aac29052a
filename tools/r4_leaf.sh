#!/bin/bash
# cooperative Cholesky leaf: unit tests, whole suite, A/B of small-model and rank-share timings
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r4_leaf; rm -rf $O; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_units.py -q -x -k "cholesky or trinv or positive_definite" 2>&1 | tail -3 || exit 1
for v in 0 1 0 1; do
  DGP_LEAF_COOP=$v DGP_BENCH_PATH=partial timeout -k 10 200 python bench.py --N 12496 --steps 40 --warmup 5 --no-cpu-baseline --nat-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DGP_LEAF_COOP=$v N/8 three-stage', round(d['ms_per_step_median'],3), 'ms', {k:round(x,2) for k,x in d['breakdown_ms_per_step'].items()})" | tee -a $O/ab.txt
done
for v in 0 1; do
  DGP_LEAF_COOP=$v timeout -k 10 200 python bench.py --steps 20 --no-cpu-baseline --nat-steps 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DGP_LEAF_COOP=$v N', round(d['ms_per_step_median'],3), 'ms; nat-adam', round(d['nat_adam_ms_per_iteration'],2))" | tee -a $O/ab.txt
done
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -40 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
