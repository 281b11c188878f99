# A/B of library variants on the one-lane closed loop (GPU box): bash tools/ab_loop_lane.sh base <variant> ...
for a in "--workload two_mass --op loop" "--workload msd --op loop"; do
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
      timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'], round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['check']['checksum'])" || exit 1
    done
  done
done
