#!/bin/bash
# PMC passes over tools/wide_bench and tools/gram_bench (per-kernel wait / issue / instruction-fetch counters).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/pmc_wide
rm -rf $O; mkdir -p $O
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_CYCLES_SALU" "SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/w$i -- $R/tools/wide_bench 4096 1000064 > /dev/null 2>&1 || { echo "wide pass $i failed"; exit 1; }
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/g$i -- $R/tools/gram_bench 70000 1000000 > /dev/null 2>&1 || { echo "gram pass $i failed"; exit 1; }
done
echo ok
