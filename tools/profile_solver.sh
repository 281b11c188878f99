#!/bin/bash
# Kernel trace of the gain runs on the two configurations the blocked solver serves (long horizon, quadrotor).
# Usage (on the GPU box, from the repository root): bash tools/profile_solver.sh <tag>
set -e
tag=${1:-solver}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for wl in long quadrotor; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_$wl -- python3 $root/bench.py --op gain --workload $wl --steps 2 --warmup 1 --no-cpu-baseline --no-also > $root/gpurun_out/${tag}_$wl.log 2>&1
  f=$(ls -t $(find $root/gpurun_out/${tag}_$wl -name "*kernel_stats.csv") | head -1)
  cp $f $root/gpurun_out/${tag}_${wl}_kernel_stats.csv
  python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'lu_' in r['Name']:
        print(r['Name'].split('::')[-1][:60], r['Calls'], round(float(r['AverageNs']) / 1e6, 3), 'ms')
PY
done
