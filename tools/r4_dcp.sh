#!/bin/bash
# dC row-panel kernel: layer-product unit tests (every element vs NumPy), isolated timing vs gemm_tallu, then the iteration
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_dcp; mkdir -p $O; cd $R
timeout -k 10 400 python -m pytest tests/test_gpu_units.py -q -x -k "layer_products" > $O/pytest_lp.log 2>&1 || { echo "layer products failed"; tail -30 $O/pytest_lp.log; exit 1; }
tail -2 $O/pytest_lp.log
timeout -k 10 60 tools/tall_bench 1000000 10 8 72 | tee $O/t.txt || exit 1
timeout -k 10 60 tools/tall_bench 1000000 10 1 72 | tee -a $O/t.txt || exit 1
for v in 0 1; do
  DGP_DCPANEL=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 --nat-steps 0 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DGP_DCPANEL=$v', round(d['ms_per_step_median'],2), 'ms; frac', round(d['roofline']['frac'],3), 'elbo', d['elbo_last'])" | tee -a $O/ab.txt || exit 1
done
if [ "$DCP_PMC" = "1" ]; then
  cd /tmp
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- $R/tools/tall_bench 1000000 3 8 72 > /dev/null 2>&1
  cd $R; python3 tools/pmc_table.py $O/pmc | grep -E "kernel|dc_panel|tallu" | tee $O/pmc_table.txt
fi
