cd "$GRAFT_REPO_ROOT"
for round in 1 2; do
  for a in "$@"; do
    echo -n "$a "
    LRM_LIB_PATH=$PWD/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$a.so timeout -k 10 120 python bench.py --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('fused %.4f far %.4f reachable %.4f' % (d['roofline']['kernel_ms'], k['fused_all_unreachable']['tol']['kernel_ms'], k['fused_all_reachable']['tol']['kernel_ms']))"
  done
done
