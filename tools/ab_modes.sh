# Same-box A/B of per-kernel durations (rocprofv3 kernel trace of bench.py): the ride-along forms on (du inside the Gram
# launch of G_d, mean0 inside the Ct launch) against off, twice each.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for v in "A 1 1" "B 0 0" "C 1 1" "D 0 0"; do
  set -- $v
  export DGP_DU_IN_GRAM=$2 DGP_WIDE_MEAN=$3
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ab_$1 -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 || exit 1
  echo "== du_in_gram=$2 wide_mean=$3"
  python3 $R/tools/kstats.py $(find $R/gpurun_out/ab_$1 -name "*kernel_trace.csv" | head -1) gram_kernel wide_kernel "128, 16, 16"
done
