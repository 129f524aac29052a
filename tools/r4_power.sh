#!/bin/bash
# Is the fp64 matrix pipeline power-limited?  Board power, clocks and temperatures sampled with rocm-smi while one kernel runs back to back
# for several seconds (tools/tall_bench, 10^6 rows, D = 8; tools/mfma_power, tools/copy_power: build them with hipcc --offload-arch=gfx950 -O3
# first), and while bench.py runs its iterations.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; O=gpurun_out/r4_power; mkdir -p $O
sample() { for i in $(seq 1 ${1:-6}); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power \(W\)|sclk|Temperature \(Sensor (junction|memory)" | tr -s ' \t' ' ' | tr '\n' ';'; echo; sleep 0.4; done; }
echo "== idle"; sample 3
if [ -x tools/mfma_power ]; then
echo "== MFMA_registers_only (tools/mfma_power 4)"
timeout -k 10 60 tools/mfma_power 4 > $O/run_mfma.txt 2>&1 &
pid=$!; sleep 1.5; sample 5; wait $pid; tail -1 $O/run_mfma.txt
echo "== MFMA_registers_only_20s (tools/mfma_power 20, samples from second 12 on)"
timeout -k 10 60 tools/mfma_power 20 > $O/run_mfma20.txt 2>&1 &
pid=$!; sleep 12; sample 8; wait $pid; tail -1 $O/run_mfma20.txt
fi
if [ -x tools/copy_power ]; then
echo "== HBM_copy (tools/copy_power 4)"
timeout -k 10 60 tools/copy_power 4 > $O/run_copy.txt 2>&1 &
pid=$!; sleep 1.5; sample 5; wait $pid; tail -1 $O/run_copy.txt
fi
for v in "16 Gram" "4 T_row_sums_only" "1 T_with_stores" "8 dC_tall_tile" "32 g_row_panel" "1024 Ct_solve"; do
  set -- $v
  reps=400; [ $1 = 32 -o $1 = 1024 ] && reps=2500        # (1.5 - 1.8 ms launches: enough of them to sample beside)
  echo "== $2 (tools/tall_bench 1000000 $reps 8 $1)"
  timeout -k 10 120 tools/tall_bench 1000000 $reps 8 $1 > $O/run_$2.txt 2>&1 &
  pid=$!; sleep 1.5; sample 5; wait $pid; tail -1 $O/run_$2.txt
done
if [ -x tools/tall_bench_SPREAD ]; then
echo "== T_with_stores_spread_form (tools/tall_bench_SPREAD 1000000 400 8 1)"
timeout -k 10 120 tools/tall_bench_SPREAD 1000000 400 8 1 > $O/run_spread.txt 2>&1 &
pid=$!; sleep 1.5; sample 5; wait $pid; tail -1 $O/run_spread.txt
fi
echo "== bench.py --steps 150 --warmup 5 --nat-steps 0 --no-cpu-baseline (whole iterations)"
timeout -k 10 300 python bench.py --steps 150 --warmup 5 --nat-steps 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err &
pid=$!; sleep 9; sample 8; wait $pid; python -c "
import json; d=json.load(open('$O/bench.json')); print('bench: ms_per_step', round(d['ms_per_step'],2), 'roofline.frac', round(d['roofline']['frac'],3))"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power (W)"
