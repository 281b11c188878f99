# HBM traffic of the benched kernels from the PMC counters: one rocprofv3 --pmc pass per counter and workload.
# Usage (on the GPU box): bash tools/pmc_traffic.sh ; then python tools/pmc_summarize.py here.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
mkdir -p $O
pass() { name=$1; ctr=$2; shift 2; timeout -s KILL 250 rocprofv3 --pmc $ctr --output-format csv -d $O/${name}_$ctr -o out -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $O/${name}_$ctr.json 2> $O/${name}_$ctr.err; }
for ctr in WRITE_SIZE FETCH_SIZE; do
  pass qp_quadrotor $ctr --workload quadrotor
  pass qp_long $ctr --workload long
  pass loop_two_mass $ctr --op loop
  pass rollout_msd $ctr --workload msd --op rollout --batch 1048576
  pass gain_long $ctr --workload long --op gain
done
find $O -name "*counter_collection.csv" | wc -l
