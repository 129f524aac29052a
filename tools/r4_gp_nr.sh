#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_gp_nr; mkdir -p $O; cd $R
timeout -k 10 60 tools/tall_bench 1000000 10 8 32 | tail -1 | sed 's/^/full kernel:      /' | tee $O/t.txt
timeout -k 10 60 tools/tall_bench_NR 1000000 10 8 32 | tail -1 | sed 's/^/k-tiles only:     /' | tee -a $O/t.txt
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- $R/tools/tall_bench_NR 1000000 3 8 32 > /dev/null 2>&1
cd $R; python3 tools/pmc_table.py $O/pmc g_panel | tee -a $O/t.txt
