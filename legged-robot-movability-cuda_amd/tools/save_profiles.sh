#!/bin/bash
# copies the r03 profile set from gpurun_out/ into profiles/ (gpurun merges every call's files into the same directories: newest file of each kind)
set -e
cd /root/repo
newest() { ls -t $1 | head -1; }
o=gpurun_out/r03
cp $o/r03_hbm_traffic.json $o/r03_valu.json profiles/
cp $o/bench.json profiles/r03_bench.json
cp $o/bench_under_rocprof.json profiles/r03_bench_under_rocprof.json
cp $(newest "$o/stats/*/*kernel_stats.csv") profiles/r03_kernel_stats.csv
cp $(newest "$o/pmc/fetch/*/*counter_collection.csv") profiles/r03_pmc_fetch_size.csv
cp $(newest "$o/pmc/write/*/*counter_collection.csv") profiles/r03_pmc_write_size.csv
cp $(newest "$o/pmc/sq/*/*counter_collection.csv") profiles/r03_pmc_sq.csv
cp gpurun_out/r03_c3/r03_c3_evidence.json profiles/
cp $(newest "gpurun_out/r03_c3/stats/*/*kernel_stats.csv") profiles/r03_c3_kernel_stats.csv
cat $(newest "gpurun_out/r03_c3/pmc_sq/*/*counter_collection.csv") $(newest "gpurun_out/r03_c3/pmc_sq2/*/*counter_collection.csv") > profiles/r03_c3_pmc_sq.csv
cp gpurun_out/r03_pytest_gpu.log profiles/r03_pytest_gpu.log
ls -la profiles | grep r03 | awk '{print $5, $9}'
