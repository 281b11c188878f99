# A/B of library variants on the blocked solver (GPU box): bash tools/ab_gain.sh "<bench args>" base <variant> ...
args=$1; shift
for lib in "$@"; do
  if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
  echo "== $lib"
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 $args | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"], d[\"config\"][\"op\"], round(d[\"ms_per_step\"],3), d[\"check\"])" || exit 1
done
