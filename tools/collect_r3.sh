#!/bin/bash
# Round-3 evidence run on the GPU box: tests, bench (with CPU baseline), smoke, rocprofv3 kernel stats, PMC traffic
# passes. Stops at the first failing GPU step. Post-process with tools/summarize_r3.py (writes profiles/r3_*).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r3_final
rm -rf $O; mkdir -p $O
echo ${COMMIT:-unknown} > $O/commit.txt
cd $R
timeout -k 10 500 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || { echo "pmc mfma failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || { echo "pmc lds failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/stats_bench.json 2>/dev/null || { echo "stats failed"; exit 1; }
cd $R
python tools/summarize_r3.py $O $O/summary || { echo "summarize failed"; exit 1; }
python tools/iter_timeline.py $(find $O/stats -name '*kernel_trace.csv' | head -1) > $O/summary/r3_iteration_timeline.txt 2>&1 || echo "timeline failed"
# the bench line proper reads the traffic figure measured just above
DGP_TRAFFIC_JSON=$O/summary/r3_pmc_traffic.json timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
timeout -k 10 400 python tools/configs_check.py > $O/summary/r3_other_configs.txt 2>&1 || { echo "configs_check failed"; tail -5 $O/summary/r3_other_configs.txt; exit 1; }
cat $O/summary/r3_other_configs.txt
