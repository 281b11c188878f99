# SQ counters of the quadrotor rollout kernel (two rocprofv3 --pmc passes).  Usage (GPU box): bash tools/pmc_rollout.sh
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_ro; mkdir -p $O
A="--no-cpu-baseline --no-also --workload quadrotor --op rollout --steps 3 --warmup 1"
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -o p1 -- python3 $R/bench.py $A > $O/p1.json 2> $O/p1.err
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/p2 -o p2 -- python3 $R/bench.py $A > $O/p2.json 2> $O/p2.err
find $O -name "*counter_collection.csv"
