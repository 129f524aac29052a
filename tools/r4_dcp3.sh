#!/bin/bash
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r4_dcp3; rm -rf $O; mkdir -p $O
for v in 0 1; do
  cd /tmp
  DGP_DCPANEL=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr$v -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --nat-steps 0 > /dev/null 2>&1
  cd $R
  echo "== DGP_DCPANEL=$v" | tee -a $O/t.txt
  python3 tools/iter_timeline.py $(find $O/tr$v -name '*kernel_trace.csv' | head -1) 800 | grep -E "iteration|tall_kernel|tallu|dc_panel" | tee -a $O/t.txt
done
for v in 0 1 0 1; do
  DGP_DCPANEL=$v timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --nat-steps 0 2> /dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('DGP_DCPANEL=$v', round(d['ms_per_step_median'],2))" | tee -a $O/t.txt
done
