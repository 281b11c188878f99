set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof2
mkdir -p $O
run() { name=$1; shift; timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline "$@" > $O/$name.json 2> $O/$name.err; }
run qp_two_mass
run qp_quadrotor --workload quadrotor --steps 5 --warmup 1
run qp_long --workload long --steps 5 --warmup 1
run loop_two_mass --op loop --steps 5 --warmup 1
run gain_two_mass --op gain --steps 5 --warmup 1
run gain_long --workload long --op gain --steps 2 --warmup 1
run rollout_msd --workload msd --op rollout --batch 1048576 --steps 5 --warmup 1
find $O -name "*kernel_stats.csv" | head -20
