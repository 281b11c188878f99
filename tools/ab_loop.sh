# A/B of library variants on the closed loop (GPU box): bash tools/ab_loop.sh   (variants: build/exp/libmpcx_<name>.so from tools/build_variant.sh, e.g. noaffine = closed_loop.hip -DMPCX_NO_AFFINE; "base" is the in-tree library)
for lib in noaffine base; do
  if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$PWD/build/exp/libmpcx_$lib.so; fi
  echo "== $lib"
  for a in "--workload quadrotor" "--workload long" "--workload two_mass --interleave 1 --batch 65536" "--workload msd --interleave 1" "--workload quadrotor --f 10 --v 10" "--workload long --f 40 --v 40"; do
    timeout -k 10 100 python bench.py --no-cpu-baseline --no-also --steps 3 --warmup 1 --op loop $a | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d[\"config\"][\"workload\"], d[\"config\"].get(\"batch_per_gpu\"), round(d[\"ms_per_step\"],4))" || exit 1
  done
done
