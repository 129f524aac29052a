#!/bin/bash
# Round-4 attribution run: new parity tests, then the tall-tile kernels with / without their stores,
# event-timed and under PMC (clock = GRBM_GUI_ACTIVE / duration, issue share = MFMA busy cycles / active cycles).
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_attrib
rm -rf $O; mkdir -p $O
cd $R
timeout -k 10 300 python -m pytest tests -q -m gpu -k "bo_notebook or config1_stated or layer_api or one_rocm_stack or notebook_known" > $O/pytest_new.log 2>&1 || { echo "pytest failed (continuing)"; tail -40 $O/pytest_new.log; }
tail -2 $O/pytest_new.log
timeout -k 10 120 tools/tall_bench 1000000 10 8 31 > $O/tall_bench_events.txt 2>&1 || { echo "tall_bench failed"; tail $O/tall_bench_events.txt; exit 1; }
cat $O/tall_bench_events.txt
cd /tmp
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_tall -- $R/tools/tall_bench 1000000 3 8 31 > $O/pmc_tall.log 2>&1 || { echo "pmc tall failed"; tail $O/pmc_tall.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_tall2 -- $R/tools/tall_bench 1000000 3 8 31 > $O/pmc_tall2.log 2>&1 || echo "pmc tall2 failed (continuing)"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { echo "bench under rocprofv3 failed"; tail -5 $O/bench_under_rocprof.err; exit 1; }
cat $O/bench_under_rocprof.json
cd $R
python3 tools/pmc_table.py $O/pmc_tall > $O/pmc_tall_table.txt 2>&1; cat $O/pmc_tall_table.txt
python3 tools/pmc_table.py $O/pmc_tall2 > $O/pmc_tall2_table.txt 2>&1; cat $O/pmc_tall2_table.txt
