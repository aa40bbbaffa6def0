#!/bin/bash
# tools/ab_clouds.sh NAME...: the tolerance mode of library variants on the config-2 cube AND on the planar bench grid (7 % of its points in doubt)
cd "$(dirname "$0")/../.."
for a in "$@"; do
  for c in cube grid; do
    echo -n "$a $c "
    LRM_LIB_PATH=$PWD/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so python legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol --cloud $c --reps 200 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4f ms' % d['tol']['ms_per_call'], d.get('tol_check', ''))"
  done
done
