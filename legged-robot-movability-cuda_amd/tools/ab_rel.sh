#!/bin/bash
# LRM_MODE_TOL_REL of library variants on the config-2 cube: tools/ab_rel.sh NAME... (two rounds)
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for a in "$@"; do
    echo -n "$a "
    LRM_LIB_PATH=$PWD/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so timeout -k 10 120 python legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol_rel,tol --reps 200 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tol_rel %.4f ms  tol %.4f ms' % (d['tol_rel']['ms_per_call'], d['tol']['ms_per_call']))"
  done
done
