#!/bin/bash
# tolerance and bit-exact mode over cloud sizes (config-2 cube) and kinds of cloud (1e7 points): ms per call, the check against
# the bit-exact mode, [points, queued, overflowed segments].  From the repository root on the GPU box.
cd "$(dirname "$0")/../.."
line() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tol %.4f ms fast %.4f ms' % (d['tol']['ms_per_call'], d['fast']['ms_per_call']), d.get('tol_check', ''), d.get('tol_queue_counts', ''))"; }
for n in 100000000 50000000 25000000 12500000 1000000 300000 100000; do
  echo -n "cube $n "
  timeout -k 10 200 python legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol,fast --points $n --reps 100 2>/dev/null | line
done
for c in grid reachable far; do
  echo -n "$c 10000000 "
  timeout -k 10 200 python legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol,fast --cloud $c --reps 200 2>/dev/null | line
done
