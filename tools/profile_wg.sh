# Kernel-trace summaries and HBM-traffic counters of the workgroup QP kernel (quadrotor, long-horizon workloads).
# Usage (on the GPU box): bash tools/profile_wg.sh ; then python tools/pmc_summarize.py here.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
P=$R/gpurun_out/pmc
mkdir -p $O $P
run() { name=$1; shift; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err; }
run qp_two_mass
run qp_quadrotor --workload quadrotor --steps 5 --warmup 1
run qp_long --workload long --steps 5 --warmup 1
pass() { name=$1; ctr=$2; shift 2; timeout -s KILL 250 rocprofv3 --pmc $ctr --output-format csv -d $P/${name}_$ctr -o out -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 "$@" > $P/${name}_$ctr.json 2> $P/${name}_$ctr.err; }
for ctr in WRITE_SIZE FETCH_SIZE; do
  pass qp_quadrotor $ctr --workload quadrotor
  pass qp_long $ctr --workload long
done
find $O -name "*kernel_stats.csv"; find $P -name "*counter_collection.csv" | wc -l
