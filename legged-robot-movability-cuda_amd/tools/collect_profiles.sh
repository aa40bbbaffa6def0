#!/bin/bash
# Collects the round's profile set on the GPU box into gpurun_out/rXX/ (run from the repository root):
#   tools/collect_profiles.sh r04 [mode] [suffix]     (suffix: e.g. _fast for a second mode's set -> gpurun_out/r04_fast/)
# kernel stats of the default bench under rocprofv3, the three PMC passes, the un-profiled bench lines.
set -o pipefail
tag=${1:-r04}; mode=${2:-tol_rel}
root=${GRAFT_REPO_ROOT:-$PWD}
suf=${3:-}
out=$root/gpurun_out/$tag$suf
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 $root/bench.py --mode $mode --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
echo "stats pass done"
small="--mode $mode --steps 3 --warmup 1 --no-cpu-baseline --no-extras --precondition-ms 0"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc/fetch -- python3 $root/bench.py $small > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc/write -- python3 $root/bench.py $small > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $out/pmc/sq -- python3 $root/bench.py $small > /dev/null 2>&1
echo "pmc passes done"
cd $root
python3 legged-robot-movability-cuda_amd/tools/summarize_profiles.py $out/pmc 10000000 $mode $out/$tag

python3 bench.py --mode $mode > $out/bench.json 2> $out/bench.err
echo "bench done"
