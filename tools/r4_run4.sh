#!/bin/bash
# dC kernel window: timing at 250k / 500k rows, layer-product tests, rank shares
set -o pipefail
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_run4; rm -rf $O; mkdir -p $O; cd $R
bash tools/r4_dcp2.sh > $O/dcp_sizes.txt 2>&1; cat $O/dcp_sizes.txt
timeout -k 10 400 python -m pytest tests/test_gpu_units.py -q -x -k "layer_products" > $O/pytest_lp.log 2>&1 || { echo "layer products failed"; tail -30 $O/pytest_lp.log; exit 1; }
tail -2 $O/pytest_lp.log
bash tools/shard_sizes.sh > $O/shard_sizes.txt 2>&1; cat $O/shard_sizes.txt
