# A/B of the workgroup pairing in solve_rows_kernel (GPU box): times and the read traffic counter for both builds
set -e
R=$GRAFT_REPO_ROOT
cd $R
bash tools/ab_gain.sh "--op solve" nopair base nopair base
bash tools/ab_gain.sh "--op gain" nopair base nopair base
bash tools/ab_gain.sh "--op gain_rows" nopair base
cd /tmp && export TMPDIR=/tmp
for lib in nopair base; do
  if [ "$lib" = base ]; then unset MPCX_LIB; else export MPCX_LIB=$R/build/exp/libmpcx_$lib.so; fi
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pair_$lib -o out -- python3 $R/bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 --op solve > /dev/null 2>&1
  f=$(find $R/gpurun_out/pair_$lib -name "*counter_collection.csv" | head -1)
  echo "== $lib FETCH_SIZE (KiB) of solve_rows launches"
  grep solve_rows "$f" | awk -F, '{print $NF}' | tr -d '"' | head -3
done
