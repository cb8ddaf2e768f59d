#!/bin/bash
# Same-box A/B of Griffin-Lim library variants (tools/bin/lib_NAME.so): tools/gl_bench.py for each, ROUNDS times round-robin.
#   bash tools/gl_ab.sh ROUNDS NAME1 NAME2 ...      (extra gl_bench arguments in GL_AB_ARGS)
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for n in "$@"; do
    echo -n "$n: "
    SSTTS_HIP_LIB=$R/tools/bin/lib_$n.so timeout -k 10 120 python3 $R/tools/gl_bench.py --reps 3 $GL_AB_ARGS 2>&1 | tail -1
  done
done
