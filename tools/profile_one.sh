# Kernel trace of one bench.py configuration: tools/profile_one.sh NAME [bench.py arguments...]
# Leaves gpurun_out/prof3/NAME/..._kernel_stats.csv (per-kernel calls, total and average duration).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof3
mkdir -p $O
name=$1; shift
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o $name -- python3 $R/bench.py --no-cpu-baseline --no-also "$@" > $O/$name.json 2> $O/$name.err
f=$(find $O/$name -name "*kernel_stats.csv" | head -1)
echo "== $name"; head -12 "$f" | cut -d, -f1-5 | cut -c1-150
