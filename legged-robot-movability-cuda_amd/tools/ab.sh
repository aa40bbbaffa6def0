#!/bin/bash
# A/B timing of library variants on ONE box (box-to-box variance is ~2-4 %):
#   tools/ab.sh NAME...   runs bench.py against csrc/build/variants/liblrm_NAME.so, three rounds each.
cd "$(dirname "$0")/../.."
for round in 1 2 3; do
  for a in "$@"; do
    echo -n "$a "
    LRM_LIB_PATH=$PWD/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused %.4f dist %.4f reach %.4f' % (d['roofline']['kernel_ms'], d['kernels']['dist_only']['ms'], d['kernels']['reach_only']['ms']))"
  done
done
