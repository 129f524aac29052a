for cfg in "1016 8" "1016 4" "504 8" "2040 4" "4088 1"; do
  set -- $cfg
  for n in 12500 25000 50000 100000; do
    DGP_GEMM_GRID_MIN=$1 DGP_GEMM_TILES_PER_WG=$2 timeout -k 10 200 python bench.py --N $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gmin $1 per $2 N $n', round(d['ms_per_step'],3), round(d['breakdown_ms_per_step']['mfma_contractions'],2))" || exit 1
  done
done
