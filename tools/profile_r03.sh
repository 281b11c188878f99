# Round-3 evidence: per-kernel time (rocprofv3 --kernel-trace --stats) and HBM traffic (one --pmc WRITE_SIZE and one
# --pmc FETCH_SIZE pass, each its own run) of the benched kernels, and the matrix-core counters of the contraction.
# Usage (GPU box): bash tools/profile_r03.sh [kt|pmc|all] [case name ...]     Afterwards, here: python tools/summarize_r03.py
set -e
what=${1:-all}
shift || true
only=" $* "
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03
mkdir -p $O
B="--no-cpu-baseline --no-also"
cases=(
 "qp_two_mass|"
 "qp_msd|--workload msd"
 "qp_two_mass_f32|--f 32 --v 32 --steps 5 --warmup 1"
 "qp_two_mass_held|--f 30 --v 10 --steps 5 --warmup 1"
 "qp_quadrotor|--workload quadrotor --steps 5 --warmup 1"
 "qp_long|--workload long --steps 5 --warmup 1"
 "qp_quadrotor_held|--workload quadrotor --f 50 --v 25 --steps 12 --warmup 4"
 "qp_long_held|--workload long --f 400 --v 200 --steps 5 --warmup 1"
 "qp_long_contract|--workload long --hessian 1 --steps 5 --warmup 1"
 "rollout_msd|--workload msd --op rollout --steps 5 --warmup 1"
 "rollout_two_mass|--workload two_mass --op rollout --steps 5 --warmup 1"
 "rollout_quadrotor|--workload quadrotor --op rollout --steps 5 --warmup 1"
 "solve_two_mass|--op solve --steps 5 --warmup 1"
 "gain_two_mass|--op gain --steps 5 --warmup 1"
 "gain_rows_two_mass|--op gain_rows --steps 5 --warmup 1"
 "loop_two_mass|--op loop --steps 5 --warmup 1"
 "loop_quadrotor|--workload quadrotor --op loop --steps 3 --warmup 1"
 "gain_quadrotor|--workload quadrotor --op gain --steps 2 --warmup 1"
 "gain_rows_quadrotor|--workload quadrotor --op gain_rows --steps 2 --warmup 1"
 "gain_long|--workload long --op gain --steps 2 --warmup 1"
 "gain_rows_long|--workload long --op gain_rows --steps 2 --warmup 1"
)
for c in "${cases[@]}"; do
  name=${c%%|*}; args=${c#*|}
  if [ "$only" != "  " ] && [[ "$only" != *" $name "* ]]; then continue; fi
  if [ "$what" != "pmc" ]; then
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$name -o $name -- python3 $R/bench.py $B $args > $O/kt_$name.json 2> $O/kt_$name.err
    echo "kt $name done"
  fi
  if [ "$what" != "kt" ]; then
    short=$(echo $args | sed -e 's/--steps [0-9]* --warmup [0-9]*//')
    for ctr in WRITE_SIZE FETCH_SIZE; do
      timeout -s KILL 200 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${name}_$ctr -o out -- python3 $R/bench.py $B --steps 2 --warmup 1 $short > $O/pmc_${name}_$ctr.json 2> $O/pmc_${name}_$ctr.err
    done
    if [ "$name" = "qp_long_contract" ]; then
      timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_${name}_MFMA -o out -- python3 $R/bench.py $B --steps 3 --warmup 1 $short > $O/pmc_${name}_MFMA.json 2> $O/pmc_${name}_MFMA.err
    fi
    echo "pmc $name done"
  fi
done
find $O -name "*kernel_stats.csv" | wc -l; find $O -name "*counter_collection.csv" | wc -l
