# A/B of library variants on the quadrotor rollout (GPU box): bash tools/ab_rollout.sh base <variant> ...
for lib in "$@"; do
  if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
  echo "== $lib"
  for a in "--workload quadrotor --op rollout" "--workload long --op rollout"; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"], round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"frac\"],4))" || exit 1
  done
done
