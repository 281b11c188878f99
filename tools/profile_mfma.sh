# Kernel-trace and MFMA counters of the FP64 contraction kernel (long-horizon workload, --hessian 1).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_mfma; mkdir -p $O
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --workload long --hessian 1 --steps 5 --warmup 1 > $O/kt.json 2> $O/kt.err
timeout -s KILL 250 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc -o pmc -- python3 $R/bench.py --no-cpu-baseline --workload long --hessian 1 --steps 3 --warmup 1 > $O/pmc.json 2> $O/pmc.err
ls $O/kt $O/pmc
