#!/bin/bash
# SQ counters of the tolerance-mode kernels of library variant $1 (csrc/build/variants/liblrm_$1.so; "default" = the
# in-tree liblrm.so): separate PMC passes of tools/bench_modes.py, averages per kernel -> stdout.
cd /tmp && export TMPDIR=/tmp
if [ "$1" != "default" ]; then export LRM_LIB_PATH=$GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_$1.so; fi
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_$1/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol --reps 5 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = "main" if ("dist_tol" in k or "dist_tab" in k) else "fix" if "fixup" in k else None
        if name: acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc): print("$1", k[0], k[1], sum(acc[k])/len(acc[k]))
PY
