# Round-2 evidence: per-kernel time (rocprofv3 --kernel-trace --stats) and HBM traffic (one --pmc WRITE_SIZE and one
# --pmc FETCH_SIZE pass) of the benched kernels.  Usage (GPU box): bash tools/profile_r02.sh [kt|pmc|all] [one case name]
# Afterwards, here: python tools/summarize_r02.py
set -e
what=${1:-all}
only=${2:-}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02
mkdir -p $O
B="--no-cpu-baseline --no-also"
cases=(
 "qp_two_mass|"
 "qp_msd|--workload msd"
 "qp_quadrotor|--workload quadrotor --steps 5 --warmup 1"
 "qp_long|--workload long --steps 5 --warmup 1"
 "qp_quadrotor_held|--workload quadrotor --f 50 --v 25 --steps 12 --warmup 4"
 "qp_long_held|--workload long --f 400 --v 200 --steps 5 --warmup 1"
 "rollout_msd|--workload msd --op rollout --steps 5 --warmup 1"
 "rollout_two_mass|--workload two_mass --op rollout --steps 5 --warmup 1"
 "rollout_quadrotor|--workload quadrotor --op rollout --steps 5 --warmup 1"
 "gain_long|--workload long --op gain --steps 2 --warmup 1"
 "gain_quadrotor|--workload quadrotor --op gain --steps 2 --warmup 1"
)
for c in "${cases[@]}"; do
  name=${c%%|*}; args=${c#*|}
  if [ -n "$only" ] && [ "$only" != "$name" ]; then continue; fi
  if [ "$what" != "pmc" ]; then
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o $name -- python3 $R/bench.py $B $args > $O/kt_$name.json 2> $O/kt_$name.err
    echo "kt $name done"
  fi
  if [ "$what" != "kt" ]; then
    for ctr in WRITE_SIZE FETCH_SIZE; do
      timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -o out -- python3 $R/bench.py $B --steps 2 --warmup 1 $(echo $args | sed -e 's/--steps [0-9]* --warmup [0-9]*//') > $O/pmc_${name}_$ctr.json 2> $O/pmc_${name}_$ctr.err
    done
    echo "pmc $name done"
  fi
done
find $O -name "*kernel_stats.csv" | wc -l; find $O -name "*counter_collection.csv" | wc -l
