#!/bin/bash
# A/B timing of library variants on ONE box: tools/ab_modes.sh NAME... runs tools/bench_modes.py --modes tol against
# csrc/build/variants/liblrm_NAME.so, two rounds each.
cd "$(dirname "$0")/../.."
for round in 1 2; do
  for a in "$@"; do
    echo -n "$a "
    LRM_LIB_PATH=$PWD/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so python legged-robot-movability-cuda_amd/tools/bench_modes.py --modes ${MODES:-tol} --reps 200 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print(' '.join('%s %.4f ms' % (k, v['ms_per_call']) for k, v in d.items() if isinstance(v, dict) and 'ms_per_call' in v), d.get('tol_check', ''))"
  done
done
