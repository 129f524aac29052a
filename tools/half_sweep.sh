for n in 12500 25000 50000; do for v in 384 1024 4096 100000; do
  DGP_HALF_MIN_WG=$v timeout -k 10 200 python bench.py --N $n --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N $n min_half $v', round(d['ms_per_step'],3), round(d['breakdown_ms_per_step']['mfma_contractions'],2))" || exit 1
done; done
