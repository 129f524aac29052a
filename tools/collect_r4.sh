#!/bin/bash
# Round-4 evidence run on the GPU box: tests, smoke, rocprofv3 PMC passes (traffic, MFMA, LDS) and kernel stats of bench.py,
# then the bench line proper (reading the traffic figure measured just above, whose commit must be HEAD), the other
# configurations and the rank shares.  Stops at the first failing GPU step.  Post-processed by tools/summarize_r4.py
# (writes profiles/r4_*).   usage: COMMIT=$(git rev-parse --short HEAD) bash tools/collect_r4.sh
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4_final
rm -rf $O; mkdir -p $O
echo ${COMMIT:-unknown} > $O/commit.txt
cd $R
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1 || { echo "pytest failed"; tail -20 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { echo "smoke failed"; tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
cd /tmp
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --nat-steps 0 --breakdown-steps 2"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1 || { echo "pmc fetch failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1 || { echo "pmc write failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1 || { echo "pmc mfma failed"; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -- $B > /dev/null 2>&1 || { echo "pmc lds failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --nat-steps 0 > $O/stats_bench.json 2>/dev/null || { echo "stats failed"; exit 1; }
cd $R
python tools/summarize_r4.py $O $O/summary || { echo "summarize failed"; exit 1; }
python tools/iter_timeline.py $(find $O/stats -name '*kernel_trace.csv' | head -1) 150 3 > $O/summary/r4_iteration_timeline.txt 2>&1 || echo "timeline failed"
# the traffic figure must have been measured at the commit under test
python - <<PY || { echo "traffic json is not from this commit"; exit 1; }
import json
t = json.load(open("$O/summary/r4_pmc_traffic.json"))
want = open("$O/commit.txt").read().strip()
assert t["commit"] == want and want != "unknown", (t["commit"], want)
PY
DGP_TRAFFIC_JSON=$O/summary/r4_pmc_traffic.json timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
cp $O/bench.json $O/summary/r4_bench.json
timeout -k 10 400 python tools/configs_check.py > $O/summary/r4_other_configs.txt 2>&1 || { echo "configs_check failed"; tail -5 $O/summary/r4_other_configs.txt; exit 1; }
cat $O/summary/r4_other_configs.txt
bash tools/shard_sizes.sh > $O/summary/r4_shard_sizes.txt 2>&1 || { echo "shard sizes failed"; tail $O/summary/r4_shard_sizes.txt; exit 1; }
cat $O/summary/r4_shard_sizes.txt
cp $O/pytest_gpu.log $O/summary/r4_pytest_gpu.log; cp $O/smoke.log $O/summary/r4_smoke.log
