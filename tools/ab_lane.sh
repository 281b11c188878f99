# A/B of qp_lane variants (GPU box): bash tools/ab_lane.sh base <variant> ...   (alternating, two rounds)
for a in "--f 20 --v 20 --steps 20" "--f 32 --v 32" "--f 30 --v 10" "--f 8 --v 8" "--workload msd --f 32 --v 32 --batch 1048576"; do
  for round in 1 2; do
    for lib in "$@"; do
      if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
      timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 3 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', d['config']['workload'], round(d['ms_per_step'],4), round(d['roofline']['frac'],4), d['check']['checksum'])" || exit 1
    done
  done
done
