# A/B of library variants on the workgroup-kernel builds (GPU box): bash tools/ab_wg.sh base <variant> ...
# (variants are build/exp/libmpcx_<variant>.so from tools/build_variant.sh; "base" is the in-tree library)
cases=("--workload quadrotor --f 50 --v 25" "--workload long --f 400 --v 200" "--workload quadrotor" "--workload long")
for lib in "$@"; do
  if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
  echo "== $lib"
  for a in "${cases[@]}"; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 10 --warmup 2 $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"], round(d[\"ms_per_step\"],4), round(d[\"roofline\"][\"frac\"],4))" || exit 1
  done
done
