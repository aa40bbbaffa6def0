#!/bin/bash
# usage: prof_one.sh NAME  -> kernel stats of bench_modes tol with variant NAME
cd /tmp && export TMPDIR=/tmp
LRM_TOL_PLANE_TABLE=${TABLE:-0} LRM_TOL_DEBUG=1 LRM_LIB_PATH=$GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$1.so rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -- python3 $GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol --reps 200 2>&1 | grep -E "workgroups|points" | cut -c1-200
python3 - <<PY
import csv,glob
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/prof_$1/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)): print("$1", r["Name"][28:60], r["Calls"], r["AverageNs"])
PY
