#!/bin/bash
# Runs tools/probe_subsets.py once per variant library under build/exp (A/B experiments on the GPU box).
# usage: tools/ab_subsets.sh <workload> <batch> <variant> [<variant> ...]
wl=$1; batch=$2; shift 2
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  MPCX_LIB=$PWD/build/exp/libmpcx_$v.so timeout -k 10 120 python tools/probe_subsets.py $wl $batch 1 2>&1 | grep "W=" || exit 1
done
