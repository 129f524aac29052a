#!/bin/bash
# Round-4 run 3: the row-panel kernel (gemm_gpanel.h): unit test first (short timeout), then the whole GPU suite, then A/B.
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_run3
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 150 python -m pytest tests/test_gpu_units.py -q -x -k "g_panel" > $O/pytest_gpanel.log 2>&1 || { echo "g_panel unit test failed"; tail -40 $O/pytest_gpanel.log; exit 1; }
tail -2 $O/pytest_gpanel.log
for v in 0 1 0 1; do
  DGP_GPANEL=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --nat-steps 0 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DGP_GPANEL=$v', round(d['ms_per_step_median'],2), 'ms; frac', round(d['roofline']['frac'],3), {k:round(v,2) for k,v in d['breakdown_ms_per_step'].items()}, 'elbo', d['elbo_last'])" | tee -a $O/ab.txt || exit 1
done
timeout -k 10 700 python -m pytest tests -q -m gpu -x > $O/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -60 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
