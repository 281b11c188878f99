#!/bin/bash
# Runs bench.py once per variant library under build/exp (A/B experiments on the GPU box).
# usage: tools/ab_bench.sh "<bench args>" <variant> [<variant> ...]
args=$1; shift
for v in "$@"; do
  MPCX_LIB=$PWD/build/exp/libmpcx_$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-also $args 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('$v', d['config']['workload'], d['config']['op'], 'ms %.4f' % d['roofline']['kernel_ms'], 'frac %.3f' % d['roofline']['frac'])" || exit 1
done
