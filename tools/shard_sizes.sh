# The per-rank work of 1 / 2 / 4 / 8 ranks on ONE GPU (bench.py --N n, no collective), in the three forms a rank can run:
#   step     dgp_grad_step, no communicator (the single-process form)
#   comm1    dgp_grad_step with a one-rank library-owned RCCL communicator attached (DGP_COMM=native's path)
#   partial  dgp_grad_partial -> dgp_grad_finish (the default collective's three-stage form, the all-reduce itself omitted)
for n in 100000 50000 25008 12496; do
  for mode in step comm1 partial; do
    env=""
    [ $mode = comm1 ] && export DGP_BENCH_ONE_RANK_COMM=1 || unset DGP_BENCH_ONE_RANK_COMM
    [ $mode = partial ] && export DGP_BENCH_PATH=partial || unset DGP_BENCH_PATH
    timeout -k 10 200 python bench.py --N $n --steps 30 --warmup 5 --no-cpu-baseline --nat-steps 0 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('N', $n, '$mode', round(d['ms_per_step_median'],2), {k:round(v,2) for k,v in d['breakdown_ms_per_step'].items()})" || exit 1
  done
done
