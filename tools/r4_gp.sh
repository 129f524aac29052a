#!/bin/bash
# row-panel kernel: unit test + isolated timing (+ PMC when GP_PMC=1)
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_gp
mkdir -p $O
cd $R
timeout -k 10 150 python -m pytest tests/test_gpu_units.py -q -x -k "g_panel" 2>&1 | tail -2 || { echo "unit test failed"; exit 1; }
timeout -k 10 60 tools/tall_bench 1000000 10 8 48 | tee $O/gp_time.txt || exit 1
timeout -k 10 60 tools/tall_bench 124928 20 8 32 | tee -a $O/gp_time.txt || exit 1
if [ "$GP_PMC" = "1" ]; then
  cd /tmp
  timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- $R/tools/tall_bench 1000000 3 8 32 > /dev/null 2>&1
  timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/pmc2 -- $R/tools/tall_bench 1000000 3 8 32 > /dev/null 2>&1
  cd $R
  python3 tools/pmc_table.py $O/pmc g_panel | tee $O/pmc_table.txt
  python3 tools/pmc_table.py $O/pmc2 g_panel | tee -a $O/pmc_table.txt
fi
