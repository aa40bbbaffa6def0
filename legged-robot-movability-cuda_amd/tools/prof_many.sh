#!/bin/bash
# usage: tools/prof_many.sh NAME...  -> rocprofv3 kernel stats of bench_modes tol for each variant, appended to gpurun_out/prof_many.txt
cd "$(dirname "$0")"
for v in "$@"; do ./prof_variant.sh $v >> $GRAFT_REPO_ROOT/gpurun_out/prof_many.txt 2>&1 || exit 1; done
