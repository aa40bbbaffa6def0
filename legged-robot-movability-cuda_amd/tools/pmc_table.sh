#!/bin/bash
cd /tmp && export TMPDIR=/tmp
export LRM_TOL_PLANE_TABLE=1
export LRM_LIB_PATH=$GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/csrc/build/variants/liblrm_sep.so
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE"; do
  d=$GRAFT_REPO_ROOT/gpurun_out/pmc_table/$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $d -- python3 $GRAFT_REPO_ROOT/legged-robot-movability-cuda_amd/tools/bench_modes.py --modes tol --reps 5 > /dev/null 2>&1
done
python3 - <<PY
import csv,glob,collections
acc=collections.defaultdict(list)
for f in glob.glob("$GRAFT_REPO_ROOT/gpurun_out/pmc_table/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        name = "grid" if "tolgrid" in k else "mid" if "tol_mid" in k else "fix" if "fixup" in k else None
        if name: acc[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(acc): print(k, sum(acc[k])/len(acc[k]))
PY
