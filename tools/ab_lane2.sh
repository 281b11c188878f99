# Tuning of the run-time-horizon one-lane build (GPU box): bash tools/ab_lane2.sh base <variant> ...
for a in "--steps 30" "--f 32 --v 32" "--f 30 --v 10"; do
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
      timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'], round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['check']['checksum'])" || exit 1
    done
  done
done
