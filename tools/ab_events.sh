# A/B of the bench's event placement (GPU box): one pair of events around the K steps against a pair around every step
for op in "--op qp" "--op rollout --workload msd" "--op rollout --workload quadrotor" "--op qp --workload msd"; do
  for mode in 0 1; do
    echo "== $op per-step-events=$mode"
    MPCX_BENCH_EVENTS_PER_STEP=$mode timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 $op | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['roofline']['frac'],4))" || exit 1
  done
done
