# A/B of library variants on the one-lane rollout (GPU box): bash tools/ab_rollout_lane.sh base <variant> ...
for a in "--workload two_mass --op rollout" "--workload msd --op rollout" "--workload msd --op rollout --batch 1048576"; do
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
      timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 20 --warmup 3 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'], d['config']['batch_per_gpu'], round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['check']['checksum'])" || exit 1
    done
  done
done
