for cfg in "1016 4 4088" "1016 2 4088" "1016 8 4088" "2040 4 4088" "1016 4 8184" "504 4 2040"; do
  set -- $cfg
  for n in 12496 100000; do
    DGP_GEMM_GRID_MIN=$1 DGP_GEMM_TILES_PER_WG=$2 DGP_GEMM_GRID=$3 timeout -k 10 200 python bench.py --N $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('gmin $1 per $2 gmax $3 N $n', round(d['ms_per_step'],3), round(d['breakdown_ms_per_step']['mfma_contractions'],2))" || exit 1
  done
done
