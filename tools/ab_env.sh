# usage: bash tools/ab_env.sh VAR A B [N] — alternates two values of an environment switch on the same box
V=$1; A=$2; B=$3; N=${4:-100000}
for rep in 1 2 3; do for x in $A $B; do
  env $V=$x timeout -k 10 200 python bench.py --N $N --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', round(d['ms_per_step'],3), round(d['breakdown_ms_per_step']['mfma_contractions'],2))" || exit 1
done; done
