#!/bin/bash
# copies the round's profile set (tools/collect_profiles.sh TAG MODE on the GPU box) from gpurun_out/ into profiles/
# (gpurun merges every call's files into the same directories: the newest file of each kind is taken):
#   tools/save_profiles.sh r04 [suffix]      suffix "" for the headline mode, e.g. "_fast" for another mode's set
set -e
cd "$(dirname "$0")/../.."
tag=${1:-r04}; suf=${2:-}
newest() { ls -t $1 | head -1; }
o=gpurun_out/$tag$suf
cp $o/${tag}_hbm_traffic.json profiles/${tag}${suf}_hbm_traffic.json
cp $o/${tag}_valu.json profiles/${tag}${suf}_valu.json
cp $o/bench.json profiles/${tag}${suf}_bench.json
cp $o/bench_under_rocprof.json profiles/${tag}${suf}_bench_under_rocprof.json
cp $(newest "$o/stats/*/*kernel_stats.csv") profiles/${tag}${suf}_kernel_stats.csv
cp $(newest "$o/pmc/fetch/*/*counter_collection.csv") profiles/${tag}${suf}_pmc_fetch_size.csv
cp $(newest "$o/pmc/write/*/*counter_collection.csv") profiles/${tag}${suf}_pmc_write_size.csv
cp $(newest "$o/pmc/sq/*/*counter_collection.csv") profiles/${tag}${suf}_pmc_sq.csv
ls -la profiles | grep ${tag}${suf}_ | awk '{print $5, $9}'
